#!/usr/bin/env python3
"""Throughput of the canonical-class kernels' instantiated bands on the C2-shaped workload (1024 x 2000, Kaiser(128, 0.5), stack):
[25, 200] Hz at fs = 1000 (rows 4..25) and [25, 400] Hz at fs = 2000 (rows 2..25), plus a band of the general kernels beside them."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import FSST, synth
w = synth.kaiser_window(128, 0.5)
for fs, band in ((1000, (25, 200)), (2000, (25, 400)), (1000, (25, 180))):
    X = torch.from_numpy(synth.pcg_windows(1024, 2000, fs=fs)).cuda()
    tf = FSST(fs, w, truncate_freq=band, stack=True)
    klo, K = tf.band()
    out = torch.empty((1024, 2000, 2 * K), dtype=torch.float32, device="cuda")
    for _ in range(20): tf.batch(X, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): tf.batch(X, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    gb = 1024 * 2000 * (4 + 8 * K) / 1e9
    print(f"fs {fs} band {band}: rows {klo}..{klo + K - 1}, {dt * 1e3:.4f} ms per 1024 windows ({1024 / dt / 1e6:.2f} M windows/s), "
          f"{gb / dt / 8000 * 100:.1f} % of 8 TB/s on {gb * 1e3:.1f} MB   [{tf.last_kernel()}]")
