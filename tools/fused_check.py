#!/usr/bin/env python3
"""Stress check of the fused z-score (ticket + release/acquire) against the two-pass path: the two
must be BIT-identical, over many launches, batch sizes and under uneven load."""
import os, subprocess, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from heart_sounds_segmentation_amd import FSST, synth
    w = synth.kaiser_window(128, 0.5)
    tf = FSST(1000, w, truncate_freq=(25, 200), stack=True)
    outs = []
    for B, n in ((1024, 2000), (7, 2000), (300, 1999), (64, 35500), (1, 64), (33, 129)):
        X = torch.from_numpy(synth.pcg_windows(B, n, seed=B + n)).cuda()
        for rep in range(int(sys.argv[3])):
            y = tf.batch(X)
            outs.append(y.cpu().numpy().copy())
    np.savez(sys.argv[2], *outs)
    sys.exit(0)
reps = "6"
for mode, path in (("0", "/tmp/z_two.npz"), ("1", "/tmp/z_fused.npz")):
    env = dict(os.environ, HSSFSST_FUSED_ZSCORE=mode)
    subprocess.run([sys.executable, __file__, "child", path, reps], check=True, env=env)
a, b = np.load("/tmp/z_two.npz"), np.load("/tmp/z_fused.npz")
bad = 0
for k in a.files:
    if not np.array_equal(a[k], b[k], equal_nan=True):
        bad += 1
        print("MISMATCH", k, a[k].shape, np.nanmax(np.abs(a[k] - b[k])))
print("arrays", len(a.files), "mismatching", bad)
sys.exit(1 if bad else 0)
