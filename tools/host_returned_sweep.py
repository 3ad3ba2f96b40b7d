#!/usr/bin/env python3
"""Host-returned corpus building (features back in pinned host memory) against the launch size: how close to the D2H link rate
(tools/d2h_bw.py) does CorpusBuilder get?  usage: host_returned_sweep.py [recordings=198]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import FSST, synth, corpus
nrec = int(sys.argv[1]) if len(sys.argv) > 1 else 198
tf = FSST(1000, synth.kaiser_window(128, 0.5), truncate_freq=(25, 200), stack=True)
base = [synth.recording(35500, seed=s) for s in range(8)]
hrecs = [(torch.from_numpy(np.roll(base[i % 8], 97 * i)), None) for i in range(nrec)]
for wpl in (4096, 2048, 1024, 512):
    b = corpus.CorpusBuilder(tf, device=torch.device("cuda", 0), windows_per_launch=wpl)
    first = b.build(hrecs, keep_on_device=False)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        items = b.build(hrecs, keep_on_device=False, out=first.features)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    n = len(items)
    print(f"windows_per_launch {wpl:5d}: {n} windows in {best * 1e3:.1f} ms = {n / best / 1e3:.1f} k windows/s = {n * 352000 / best / 1e9:.1f} GB/s")
    del items, first, b
