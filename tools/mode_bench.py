#!/usr/bin/env python3
"""Throughput of the three output modes on the C2-shaped workload (1024 x 2000, Kaiser(128, 0.5), band [25,200] Hz)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import FSST, synth
X = torch.from_numpy(synth.pcg_windows(1024, 2000)).cuda()
w = synth.kaiser_window(128, 0.5)
for name, kw in (("stack", dict(stack=True)), ("abs", dict(abs=True)), ("raw", dict())):
    tf = FSST(1000, w, truncate_freq=(25, 200), **kw)
    for _ in range(300): y = tf.batch(X)
    tf.set_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): y = tf.batch(X)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
    core, rest, cnt = tf.timing(); tf.set_timing(False)
    print(f"{name:6s}: {dt * 1e3:7.3f} ms per 1024 windows ({1024 / dt / 1e6:5.2f} M windows/s); core kernel {core / cnt:7.3f} ms, rest {rest / cnt:6.3f} ms; out {tuple(y.shape)} {y.dtype}")
