#!/usr/bin/env python3
"""The host-side wait of a one-window call looks at a word the kernel's last wave stores to pinned host memory instead of synchronising the stream
(hssfsst.hip, exec_impl): every result of N back-to-back drop-in calls on DIFFERENT frames is compared bit for bit with the batched device path.
A result handed out before all its features had landed would show here as a mismatch (the buffer holds an older call's features).
usage: flag_stress.py [calls=20000] [keep=8]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import FSST, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
keep = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.set_num_threads(1)
tf = FSST(1000, synth.kaiser_window(128, 0.5), truncate_freq=(25, 200), stack=True)
B = 512
X = torch.from_numpy(np.concatenate([synth.pcg_windows(B // 2, 2000, seed=5), synth.noise_windows(B // 2, 2000, seed=6)]).astype(np.float32))
ref = tf.batch(X.cuda()).cpu()                      # (B, 2000, 44): the same kernels on a device-resident batch (bit-identical by the suite's tests)
frames = [X[i].reshape(2000, 1).contiguous() for i in range(B)]
bad, held, t0 = 0, [], time.perf_counter()
rng = np.random.default_rng(1)
order = rng.integers(0, B, size=N)
for k in range(N):
    i = int(order[k])
    y = tf(frames[i])
    if not torch.equal(y, ref[i]):
        bad += 1
        if bad <= 5:
            d = (y - ref[i]).abs()
            print(f"call {k}: frame {i} differs in {(d > 0).sum().item()} of {d.numel()} features, first at flat index {int((d.reshape(-1) > 0).nonzero()[0])}")
    held.append(y)                                  # keep a few results alive: the pool hands out different buffers
    if len(held) > keep: held.pop(0)
dt = time.perf_counter() - t0
print(f"{N} calls, {bad} results differ from the batched device path; {dt / N * 1e6:.1f} us per call including the comparison; "
      f"team launches given up: {tf.fallbacks() if hasattr(tf, 'fallbacks') else 'n/a'}")
sys.exit(1 if bad else 0)
