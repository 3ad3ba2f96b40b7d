#!/usr/bin/env python3
"""Throughput of every supported window length on the C2-shaped workload (1024 x 2000, band [25,200] Hz, stack)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import FSST, synth
from scipy.signal import get_window
X = torch.from_numpy(synth.pcg_windows(1024, 2000)).cuda()
for nwin in [int(v) for v in os.environ.get("NWINS", "32,64,128,256,512").split(",")]:
    tf = FSST(1000, get_window(("kaiser", 0.5), nwin, fftbins=False), truncate_freq=(25, 200), stack=True)
    klo, K = tf.band()
    out = torch.empty((1024, 2000, 2 * K), dtype=torch.float32, device="cuda")
    for _ in range(20): tf.batch(X, out=out)
    tf.set_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): tf.batch(X, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    core, rest, cnt = tf.timing(); tf.set_timing(False)
    print(f"nwin {nwin:4d}: K = {K:3d} kept rows, {dt * 1e3:7.3f} ms per 1024 windows ({1024 / dt / 1e6:5.2f} M windows/s); core {core / cnt:7.3f} ms, z-score {rest / cnt:6.3f} ms")
