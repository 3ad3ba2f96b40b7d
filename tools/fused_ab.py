#!/usr/bin/env python3
"""Fused z-score kernel vs the two-kernel path: bit equality and timing (GPU box).
usage: fused_ab.py [--n 2000] [--batch 1024] [--steps 300]
The two-pass reference runs in a child process with HSSFSST_NO_FUSED=1 (the switch is read once per process)."""
import argparse, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from heart_sounds_segmentation_amd import FSST, synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2000)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--child", default="")
a = ap.parse_args()
w = synth.kaiser_window(128, 0.5)
X = torch.from_numpy(synth.pcg_windows(a.batch, a.n, seed=5)).cuda()
tf = FSST(1000, w, truncate_freq=(25, 200), stack=True)
out = torch.empty((a.batch, a.n, 44), dtype=torch.float32, device="cuda")
tf.batch(X, out=out)
fused = tf.check()
for _ in range(200):
    tf.batch(X, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    tf.batch(X, out=out)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / a.steps * 1e3
tf.check()
res = out.cpu().numpy()
if a.child:
    np.save(a.child, res)
    print(f"child fused={fused} {ms:.4f} ms/step")
    sys.exit(0)
print(f"n={a.n} batch={a.batch}: fused={fused} {ms:.4f} ms/step = {a.batch / ms * 1e3 / 1e6:.3f} M windows/s")
tmp = f"/tmp/fused_ab_{os.getpid()}.npy"
env = dict(os.environ, HSSFSST_NO_FUSED="1")
r = subprocess.run([sys.executable, __file__, "--n", str(a.n), "--batch", str(a.batch), "--steps", str(a.steps), "--child", tmp],
                   env=env, capture_output=True, text=True)
print(r.stdout.strip(), r.stderr.strip()[-300:])
ref = np.load(tmp)
same = np.array_equal(res, ref, equal_nan=True)
print("bit-identical to two-pass:", same, "max abs diff", float(np.nanmax(np.abs(res - ref))))
