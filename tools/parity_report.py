#!/usr/bin/env python3
"""Parity census on the GPU box: HIP path vs the fp64 oracle on a larger sample than the tests use.
Reports per input family: max relative error on robust columns, number of rounding-fragile columns,
and how many of those actually differ (an fp32-vs-fp64 rounding flip)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from heart_sounds_segmentation_amd import FSST, synth
from tests import parity
from scipy.signal import get_window

def census(name, X, fs, w, band, mode):
    kw = dict(stack=(mode == "stack"), abs=(mode == "abs"))
    got = FSST(fs, w, truncate_freq=band, **kw).batch(torch.from_numpy(X).cuda()).cpu().numpy()
    ref, hd = oracle.features(X, fs, w, band, mode, nthreads=os.cpu_count(), return_halfdist=True)
    tax = 1 if mode == "raw" else 0
    worst, frag, flipped, cols = 0.0, 0, 0, 0
    for b in range(X.shape[0]):
        o = np.moveaxis(got[b], tax, 0).reshape(got[b].shape[tax], -1)
        r = np.moveaxis(ref[b], tax, 0).reshape(ref[b].shape[tax], -1)
        scale = np.abs(r).max()
        err = np.abs(o - r).max(axis=1) / scale
        fr = hd[b] < parity.FRAG_EPS
        worst = max(worst, float(err[~fr].max()))
        frag += int(fr.sum()); flipped += int((err[fr] > parity.TOL).sum()); cols += len(fr)
        assert (err[~fr] <= parity.TOL).all(), (name, b, float(err[~fr].max()))
    return {"case": name, "signals": int(X.shape[0]), "columns": cols, "max_rel_err_robust": worst,
            "fragile_columns": frag, "fragile_columns_that_flipped": flipped}

w = synth.kaiser_window(128, 0.5)
out = [census("C2 pcg, Kaiser(128,0.5), stack", synth.pcg_windows(256, 2000), 1000, w, (25, 200), "stack"),
       census("noise N(0,1), Kaiser(128,0.5), stack", synth.noise_windows(128, 2000), 1000, w, (25, 200), "stack"),
       census("noise, Kaiser(128,0.5), raw full", synth.noise_windows(32, 2000, seed=3), 1000, w, None, "raw"),
       census("noise, Hann(128), stack", synth.noise_windows(64, 2000, seed=4), 1000, get_window("hann", 128, fftbins=False), (25, 200), "stack"),
       census("noise, Kaiser(256,10), abs (MFMA kernel, two passes)", synth.noise_windows(32, 2000, seed=5), 1000, get_window(("kaiser", 10.0), 256, fftbins=False), (25, 200), "abs"),
       census("pcg 4 kHz, Kaiser(512,0.5), stack (MFMA kernel, 32 taps)", synth.pcg_windows(32, 4000, fs=4000, seed=6), 4000, get_window(("kaiser", 0.5), 512, fftbins=False), (25, 200), "stack")]
print(json.dumps(out, indent=1))
