#!/usr/bin/env python3
"""usage: npz_equal.py a.npz b.npz -- bitwise comparison of every array"""
import sys
import numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
bad = 0
for k in a.files:
    x, y = a[k], b[k]
    same = x.shape == y.shape and np.array_equal(np.ascontiguousarray(x).view(np.uint32), np.ascontiguousarray(y).view(np.uint32))
    d = float(np.nanmax(np.abs(x.astype(np.float64) - y.astype(np.float64)))) if x.shape == y.shape else float("nan")
    print(f"{k}: {'bit-identical' if same else 'DIFFERENT'} (max abs diff {d:.3e})")
    bad += 0 if same else 1
sys.exit(1 if bad else 0)
