#!/usr/bin/env python3
"""Latency of ISOLATED launches (the GPU idles between calls, as in real-time streaming) per window length.
Kernels that use scratch memory pay for its on-demand allocation on every isolated launch on this runtime."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import FSST, synth
from scipy.signal import get_window
for nwin, fs in ((128, 1000), (256, 1000), (512, 4000), (64, 1000)):
    tf = FSST(fs, get_window(("kaiser", 0.5), nwin, fftbins=False), truncate_freq=(25, 200), stack=True)
    X = torch.from_numpy(synth.pcg_windows(64, 640, fs=fs)).cuda()
    for _ in range(5): tf.batch(X)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        time.sleep(0.02)
        t0 = time.perf_counter(); y = tf.batch(X); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    for _ in range(100): y = tf.batch(X)
    torch.cuda.synchronize(); b2b = (time.perf_counter() - t0) / 100
    print(f"nwin {nwin:4d}: isolated launch median {np.median(ts) * 1e6:7.1f} us (min {np.min(ts) * 1e6:6.1f}), back-to-back {b2b * 1e6:6.1f} us  (64 x 640 samples)")
