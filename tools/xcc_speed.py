#!/usr/bin/env python3
"""Development: how long each block (CU) of the team kernel runs, by XCD -- builds with -DHSS_T16_XCCPROBE (tools/dev.sh).
usage: xcc_speed.py lib.so [pcg|noise|zeros] [launches]   (with a -DHSS_T16_ABLATE=3 build nobody but the resolvers waits: a block's
lifetime is then its own speed)"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import synth  # noqa: E402
from tools.canon_check import load  # noqa: E402


def main():
    L = load(sys.argv[1])
    kind = sys.argv[2] if len(sys.argv) > 2 else "pcg"
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    B, n = 1024, 2000
    w = np.ascontiguousarray(synth.kaiser_window(128, 0.5))
    plan = ctypes.c_void_p()
    rc = L.hssfsst_plan_create(ctypes.byref(plan), 0, 128, w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 1000.0, 1, 25.0, 200.0, 2)
    assert rc == 0, L.hssfsst_last_error()
    xh = {"pcg": lambda: synth.pcg_windows(B, n), "noise": lambda: synth.noise_windows(B, n), "zeros": lambda: np.zeros((B, n), np.float32)}[kind]()
    X = torch.from_numpy(xh.astype(np.float32)).cuda()
    out = torch.empty((B, n, 44), dtype=torch.float32, device="cuda")
    buf = (ctypes.c_ulonglong * 1032)()
    L.hssfsst_dev_t16_xcc.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for _ in range(300):                                   # clocks settle
        L.hssfsst_exec(plan, ctypes.c_void_p(X.data_ptr()), B, n, 1, ctypes.c_void_p(out.data_ptr()), 1, None)
    rows = []
    for r in range(reps):
        rc = L.hssfsst_dev_t16_xcc(None, 1); assert rc == 0, rc
        for _ in range(20):
            L.hssfsst_exec(plan, ctypes.c_void_p(X.data_ptr()), B, n, 1, ctypes.c_void_p(out.data_ptr()), 1, None)
        rc = L.hssfsst_dev_t16_xcc(buf, 0); assert rc == 0, rc
        allw = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
        a = allw[:1024].reshape(256, 4)
        print(f"   statistics not there in time: {allw[1024]} of {20 * 128000} groups, further looks {allw[1025]}; blocked {allw[1026] / 100.0 / max(allw[1024], 1):.2f} us per miss "
              f"= {allw[1026] / 100.0 / (20 * 4096):.2f} us per wave and launch; finisher {allw[1027] / 100.0 / max(allw[1028], 1):.2f} us per signal ({allw[1028]} signals)")
        xcc, t0, t1 = a[:, 0] & 0xff, a[:, 1], a[:, 2]
        life = (t1 - t0) / 100.0                           # us (the last launch's blocks)
        start = (t0 - t0.min()) / 100.0
        end = (t1 - t0.min()) / 100.0
        rows.append((xcc, life, start, end))
        per = [life[xcc == k].mean() for k in range(8)]
        pe = [end[xcc == k].mean() for k in range(8)]
        print(f"rep {r}: lifetime by XCD (us) " + " ".join(f"{v:6.1f}" for v in per) + f" | all {life.mean():.1f} +- {life.std():.1f} (min {life.min():.1f} max {life.max():.1f})"
              f" | end by XCD " + " ".join(f"{v:6.1f}" for v in pe) + f" | starts spread {start.max():.1f}", flush=True)
    life = np.stack([r[1] for r in rows])
    xcc = rows[0][0]
    print("mean over reps, by XCD:", " ".join(f"{life[:, xcc == k].mean():6.1f}" for k in range(8)))
    print("per-CU persistence: correlation of lifetimes between consecutive reps:", " ".join(f"{np.corrcoef(life[i], life[i + 1])[0, 1]:.2f}" for i in range(len(rows) - 1)))
    by_team = life[-1].reshape(16, 16)
    print("last rep, by team (identity // 16): mean", " ".join(f"{v:6.1f}" for v in by_team.mean(1)))
    print("last rep, by team: max-min within team", " ".join(f"{v:6.1f}" for v in (by_team.max(1) - by_team.min(1))))


if __name__ == "__main__":
    main()
