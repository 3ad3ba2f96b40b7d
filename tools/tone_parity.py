import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import oracle
from heart_sounds_segmentation_amd import FSST, synth
from tests import parity
from scipy.signal import get_window
import os
fs, n = 1000.0, 2000
NW = int(os.environ.get("TP_NWIN", "128"))
t = np.arange(n) / fs
for wname in ("kaiser0.5", "hann", "blackman"):
    w = get_window(("kaiser", 0.5), NW, fftbins=False) if wname == "kaiser0.5" else get_window(wname, NW, fftbins=False)
    for f0, label in ((125.0, "on-bin"), (117.3, "off-bin"), (50.0, "50 Hz")):
        X = np.stack([np.cos(2 * np.pi * f0 * t), np.cos(2 * np.pi * f0 * t + 0.7) * (1 + 0.3 * np.sin(2 * np.pi * 3 * t)),
                      np.cos(2 * np.pi * f0 * t) + 0.5 * np.cos(2 * np.pi * 2.2 * f0 * t)]).astype(np.float32)
        for mode in ("stack", "raw"):
            tf = FSST(fs, w, truncate_freq=(25, 200), stack=(mode == "stack"))
            got = tf.batch(torch.from_numpy(X).cuda()).cpu().numpy()
            ref, hd = oracle.features(X, fs, w, (25, 200), mode, return_halfdist=True)
            worst = 0.0; msg = "ok"
            try:
                for b in range(3):
                    r = parity.check(got[b], ref[b], hd[b], 1 if mode == "raw" else 0, what=f"{wname} {label} {mode} sig{b}")
                    worst = max(worst, r["rel"])
            except AssertionError as e:
                msg = "FAIL " + str(e)[:160]
            print(f"{wname:10s} {label:8s} {mode:5s} worst rel {worst:.2e}  {msg}")
