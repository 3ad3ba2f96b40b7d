#!/usr/bin/env python3
"""Interleaved A/B of builds of libhssfsst.so on the C2 workload, timed as bench.py times a step: K execs QUEUED between two events
(no per-exec events), builds taken in turn, many rounds; prints median / min / spread of the per-exec time and the difference to the first.
usage: [AB_INPUT=pcg|noise|zeros] [AB_ROUNDS=12] [AB_STEPS=200] ab2.py lib_a.so lib_b.so ..."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import synth  # noqa: E402
from tools.ab_bench import load  # noqa: E402


def main():
    libs = sys.argv[1:]
    B = int(os.environ.get("AB_BATCH", "1024"))
    rounds = int(os.environ.get("AB_ROUNDS", "12"))
    steps = int(os.environ.get("AB_STEPS", "200"))
    kind = os.environ.get("AB_INPUT", "pcg")
    off = float(os.environ.get("AB_OFFSET", "0"))
    w = np.ascontiguousarray(synth.kaiser_window(128, 0.5))
    xh = {"pcg": lambda: synth.pcg_windows(B, 2000), "noise": lambda: synth.noise_windows(B, 2000),
          "zeros": lambda: np.zeros((B, 2000), np.float32)}[kind]().astype(np.float32) + np.float32(off)
    X = torch.from_numpy(xh).cuda()
    Ls, plans, outs = [], [], []
    for path in libs:
        L = load(path)
        plan = ctypes.c_void_p()
        rc = L.hssfsst_plan_create(ctypes.byref(plan), 0, 128, w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 1000.0, 1, 25.0, 200.0, 2)
        assert rc == 0, L.hssfsst_last_error()
        Ls.append(L); plans.append(plan); outs.append(torch.empty((B, 2000, 44), dtype=torch.float32, device="cuda"))
    xp = ctypes.c_void_p(X.data_ptr())

    def run(i, k):
        L, plan, op = Ls[i], plans[i], ctypes.c_void_p(outs[i].data_ptr())
        for _ in range(k):
            rc = L.hssfsst_exec(plan, xp, B, 2000, 1, op, 1, None)
            assert rc == 0, L.hssfsst_last_error()

    for i in range(len(libs)):
        run(i, 300)                                       # clocks settle
    torch.cuda.synchronize()
    res = [[] for _ in libs]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rd in range(rounds):
        order = list(range(len(libs)))
        if rd & 1:
            order.reverse()
        for i in order:
            run(i, 20)
            e0.record()
            run(i, steps)
            e1.record()
            e1.synchronize()
            res[i].append(e0.elapsed_time(e1) / steps)
    base = np.median(res[0])
    L0 = Ls[0]
    for i, path in enumerate(libs):
        a = np.asarray(res[i])
        L = Ls[i]
        L.hssfsst_plan_fallbacks.argtypes = [ctypes.c_void_p]
        diff = (outs[i] - outs[0]).abs().max().item()
        print(f"{os.path.basename(path):24s} median {np.median(a) * 1e3:7.2f} us  min {a.min() * 1e3:7.2f}  max {a.max() * 1e3:7.2f}   "
              f"{(np.median(a) / base - 1) * 100:+6.2f} % vs first   {360000 * B / (np.median(a) * 1e-3) / 8e12 * 100:5.2f} % of 8 TB/s   "
              f"fallbacks {L.hssfsst_plan_fallbacks(plans[i])}  maxdiff_vs_first {diff:.2e}", flush=True)


if __name__ == "__main__":
    main()
