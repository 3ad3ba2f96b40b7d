#!/usr/bin/env python3
"""Throughput of the canonical configuration (2000-sample windows, Kaiser(128, 0.5), band [25,200] Hz, stack) against the
batch size: which path a batch takes (fused single kernel / transform + z-score kernels) and what it costs."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import FSST, synth

tf = FSST(1000, synth.kaiser_window(128, 0.5), truncate_freq=(25, 200), stack=True)
Xall = torch.from_numpy(synth.pcg_windows(8192, 2000)).cuda()
out = torch.empty((8192, 2000, 44), dtype=torch.float32, device="cuda")
for _ in range(300): tf.batch(Xall[:1024], out=out[:1024])          # clocks
for B in [int(v) for v in os.environ.get("BATCHES", "1,8,33,50,128,255,256,300,512,600,768,1024,1100,1536,2048,4096,8192").split(",")]:
    X, o = Xall[:B], out[:B]
    reps = max(20, min(2000, 200000 // B))
    for _ in range(10): tf.batch(X, out=o)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): tf.batch(X, out=o)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    fused = tf.check()
    print(f"batch {B:5d}: {dt * 1e3:8.4f} ms  {B / dt / 1e6:6.3f} M windows/s  {'fused' if fused else 'two-kernel'}  "
          f"({360000 * B / dt / 8e12 * 100:5.2f} % of 8 TB/s)")
