#!/usr/bin/env python3
"""Structured (non-random) inputs against the fp64 oracle with the gate of tests/parity.py: the signals on which nearly every
cell of a frame is small against its frame's spectrum and far-moving (tones, harmonics-rich waves, impulses, steps, chirps),
under low-sidelobe windows, for every kernel (window lengths 32 ... 512, odd lengths) and several bands.
usage: adversarial_parity.py [nwin ...]   prints one line per (nwin, window, band, mode) with the worst signal; exit 1 on a miss"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from heart_sounds_segmentation_amd import FSST
from tests import parity
from scipy.signal import get_window, square, sawtooth, chirp

fs, n = 1000.0, 1536
t = np.arange(n) / fs
rng = np.random.default_rng(0)
signals = {
    "tone 125": np.cos(2 * np.pi * 125.0 * t),
    "tone 117.3": np.cos(2 * np.pi * 117.3 * t),
    "two close tones": np.cos(2 * np.pi * 100.0 * t) + 0.8 * np.cos(2 * np.pi * 104.0 * t),
    "square 40": square(2 * np.pi * 40.0 * t),
    "sawtooth 33": sawtooth(2 * np.pi * 33.0 * t),
    "impulses": (np.arange(n) % 97 == 0).astype(np.float64),
    "step": (t > 0.7).astype(np.float64) + 0.25,
    "chirp 20-400": chirp(t, 20.0, t[-1], 400.0),
    "tone + 1e-4 noise": np.cos(2 * np.pi * 60.0 * t) + 1e-4 * rng.standard_normal(n),
    "big dc + tone": 100.0 + np.cos(2 * np.pi * 80.0 * t),
}
names = list(signals)
X = np.stack([signals[k] for k in names]).astype(np.float32)
windows = {"kaiser0.5": ("kaiser", 0.5), "hann": "hann", "blackman": "blackman", "kaiser10": ("kaiser", 10.0), "flattop": "flattop"}
bands = [(25, 200), None, (300, 450)]
nwins = [int(a) for a in sys.argv[1:]] or [128]
bad = 0
for nwin in nwins:
    for wname, wspec in windows.items():
        w = get_window(wspec, nwin, fftbins=False)
        for band in bands:
            for mode in ("stack", "raw"):
                tf = FSST(fs, w, truncate_freq=band, stack=(mode == "stack"))
                if tf.band()[1] == 0:
                    continue
                got = tf.batch(torch.from_numpy(X).cuda()).cpu().numpy()
                ref, hd = oracle.features(X, fs, w, band, mode, nthreads=os.cpu_count(), return_halfdist=True)
                worst, wname_sig, fails = 0.0, "", []
                for b, nm in enumerate(names):
                    if mode == "stack" and not np.isfinite(ref[b]).all():
                        continue
                    try:
                        r = parity.check(got[b], ref[b], hd[b], 1 if mode == "raw" else 0, what=nm)
                        if r["rel"] > worst: worst, wname_sig = r["rel"], nm
                    except AssertionError as e:
                        fails.append(str(e)[:90])
                bad += len(fails)
                print(f"nwin {nwin:4d} {wname:10s} band {str(band):12s} {mode:5s} worst ok {worst:.2e} ({wname_sig})" +
                      ("".join("\n      FAIL " + f for f in fails)), flush=True)
print(f"{bad} misses")
sys.exit(1 if bad else 0)
