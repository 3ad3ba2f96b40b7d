#!/usr/bin/env python3
"""Development: where corpus.build_features spends its time (cProfile, quarter of the C3 stand-in from host memory)."""
import cProfile, pstats, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import FSST, corpus, synth
tf = FSST(1000, synth.kaiser_window(128, 0.5), truncate_freq=(25, 200), stack=True, device="cuda:0")
base = [synth.recording(35500, seed=i) for i in range(8)]
recs = [(torch.from_numpy(np.roll(base[i % 8], 97 * i)), None) for i in range(198)]
for keep in (True, False):
    corpus.build_features(recs, tf, keep_on_device=keep); torch.cuda.synchronize()
    t0 = time.perf_counter(); corpus.build_features(recs, tf, keep_on_device=keep); torch.cuda.synchronize(); print("keep", keep, "second call", time.perf_counter() - t0)
    pr = cProfile.Profile(); pr.enable(); corpus.build_features(recs, tf, keep_on_device=keep); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
