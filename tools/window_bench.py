#!/usr/bin/env python3
"""ms per 1024 x 2000 windows (stack, band [25,200]) for window families with little / heavy reassignment."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import FSST, synth
from scipy.signal import get_window
for kind, X in (("noise", torch.from_numpy(synth.noise_windows(1024, 2000, seed=1)).cuda()), ("pcg", torch.from_numpy(synth.pcg_windows(1024, 2000, seed=1)).cuda())):
  for name, w in (("kaiser(128,0.5)", synth.kaiser_window(128, 0.5)), ("hann128", get_window("hann", 128, fftbins=False)),
                ("blackman128", get_window("blackman", 128, fftbins=False)), ("kaiser(256,10)", get_window(("kaiser", 10.0), 256, fftbins=False)),
                ("kaiser(512,0.5)", get_window(("kaiser", 0.5), 512, fftbins=False)), ("hann64", get_window("hann", 64, fftbins=False))):
    tf = FSST(1000, w, truncate_freq=(25, 200), stack=True)
    out = tf.batch(X)
    for _ in range(20): tf.batch(X, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): tf.batch(X, out=out)
    torch.cuda.synchronize()
    print(f"{name:16s} {(time.perf_counter() - t0) / 50 * 1e3:8.4f} ms / 1024 windows ({kind} input)", flush=True)
