#!/usr/bin/env python3
"""What the 22-bit operands of the split-f16 window fold cost in accuracy: the canonical configuration (Kaiser(128, 0.5), band
[25, 200] Hz, stack=True) on SIGNALS x 2000 columns (default 512: ~1 M columns), the canonical kernels (csrc/fsst_canon128.hpp,
fsst_team16.hpp: fold on the f16 matrix pipe with split operands) and the general float32-fold kernels (HSSFSST_NO_CANON=1, a
child process: the switch is read when the library loads) against the float64 oracle.  Per column: max |out - ref| / max |ref| of
the signal.  usage: split_fold_census.py [signals]   (prints; tee into profiles/rNN_split_fold_census.txt)"""
import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 512


def inputs():
    from heart_sounds_segmentation_amd import synth
    return {"noise": synth.noise_windows(B, 2000, seed=77), "pcg": synth.pcg_windows(B, 2000, seed=78)}


if os.environ.get("SFC_CHILD"):
    import torch
    from heart_sounds_segmentation_amd import FSST, synth
    tf = FSST(1000, synth.kaiser_window(128, 0.5), truncate_freq=(25, 200), stack=True)
    for name, X in inputs().items():
        y = tf.batch(torch.from_numpy(X).cuda())
        np.save(os.path.join(os.environ["SFC_CHILD"], name + ".npy"), y.cpu().numpy())
    print(tf.last_kernel())
    sys.exit(0)

import oracle
from heart_sounds_segmentation_amd import synth
from tests import parity
w = synth.kaiser_window(128, 0.5)
X = inputs()
ref = {k: oracle.features(v, 1000, w, (25, 200), "stack", nthreads=os.cpu_count(), return_halfdist=True) for k, v in X.items()}
print(f"# {B} signals x 2000 columns per input kind; per-column error = max over the 44 features of |out - ref| / max |ref| of the signal")
worst = 0.0
for label, env in (("canonical kernels (split-f16 fold, 22-bit operands)", {}), ("general kernels (float32 fold), HSSFSST_NO_CANON=1", {"HSSFSST_NO_CANON": "1"})):
    with tempfile.TemporaryDirectory() as td:
        e = dict(os.environ, SFC_CHILD=td); e.update(env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(B)], env=e, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        kern = r.stdout.strip().splitlines()[-1]
        for name in X:
            got = np.load(os.path.join(td, name + ".npy"))
            rf, hd = ref[name]
            scale = np.abs(rf).max(axis=(1, 2), keepdims=True)
            err = (np.abs(got - rf) / scale).max(axis=2)          # [B][n]
            robust = hd >= parity.FRAG_EPS
            e_ = err[robust]
            if not env:
                worst = max(worst, float(e_.max()))
            print(f"{label:58s} {name:6s} columns {e_.size:8d}  max {e_.max():.3e}  99.99th pct {np.quantile(e_, 0.9999):.3e}  median {np.median(e_):.3e}  "
                  f"> 1e-6: {int((e_ > 1e-6).sum())}  > 1e-5: {int((e_ > 1e-5).sum())}  fragile {int((~robust).sum())} flipped {int((err[~robust] > parity.TOL).sum())}   [{kern}]", flush=True)
print(f"arithmetic_max_rel_err {worst:.3e}")
import bench as _bench
print(f"csrc_sha256 {_bench.csrc_digest()}")
