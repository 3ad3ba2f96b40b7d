#!/usr/bin/env python3
"""Edge shapes of the canonical configuration against the oracle: very short and ragged signals, batch 1..9, every z-score path."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from heart_sounds_segmentation_amd import FSST, synth
from tests import parity
w = synth.kaiser_window(128, 0.5)
bad = 0
for n in (1, 2, 5, 15, 16, 17, 63, 64, 65, 127, 128, 129, 191, 192, 193, 960, 961, 2047, 2048, 2049, 4097):
    for B in (1, 3, 9, 257):
        X = (np.random.default_rng(n * 1000 + B).standard_normal((B, n)) * 0.3).astype(np.float32)
        ref, hd = oracle.features(X, 1000, w, (25, 200), "stack", nthreads=8, return_halfdist=True)
        outs = {}
        for zp in ("auto", "two_launch", "team", "one_cu"):
            tf = FSST(1000, w, truncate_freq=(25, 200), stack=True)
            tf.set_zpath(zp)
            got = tf.batch(torch.from_numpy(X).cuda())
            path = tf.check()
            outs[zp] = (got, path)
            g = got.cpu().numpy()
            for b in range(B):
                if not np.isfinite(ref[b]).all():
                    continue
                try:
                    parity.check(g[b], ref[b], hd[b], 0, what=f"n={n} B={B} {zp} sig{b}")
                except AssertionError as e:
                    bad += 1
                    print("FAIL", str(e)[:160])
                    break
        base = outs["two_launch"][0]
        for zp, (got, path) in outs.items():
            same = torch.equal(torch.nan_to_num(got), torch.nan_to_num(base))
            if not same:
                bad += 1
                print(f"DIFFER n={n} B={B} {zp} (path {path}) vs two_launch")
    print(f"n={n}: paths " + " ".join(f"{zp}:{outs[zp][1]}" for zp in outs), flush=True)
print(f"{bad} problems")
