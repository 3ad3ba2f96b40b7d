#!/usr/bin/env python3
"""Latency of the drop-in FSST.__call__ with a CPU (2000, 1) tensor (the reference dataset loop's call)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import FSST, synth
tf = FSST(1000, synth.kaiser_window(128, 0.5), truncate_freq=(25, 200), stack=True)
fr = torch.from_numpy(synth.pcg_windows(1, 2000)[0]).reshape(2000, 1)
for rep in range(4):
    t0 = time.perf_counter()
    for _ in range(200): y = tf(fr)
    print(f"round {rep}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call", y.shape)
if len(sys.argv) > 1:
    rec = torch.from_numpy(synth.recording(35500)); tf(rec)
    t0 = time.perf_counter()
    for _ in range(200): y = tf(fr)
    print(f"after a 35500-sample call: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call")
if len(sys.argv) > 1:
    from heart_sounds_segmentation_amd.corpus import build_features
    build_features([(rec, None)], tf)
    t0 = time.perf_counter()
    for _ in range(200): y = tf(fr)
    print(f"after build_features: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call")
    import gc; gc.collect(); torch.cuda.empty_cache()
    t0 = time.perf_counter()
    for _ in range(200): y = tf(fr)
    print(f"after empty_cache: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call")
