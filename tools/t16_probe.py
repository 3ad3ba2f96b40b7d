#!/usr/bin/env python3
"""Development: time of the team kernels on the C2 workload and whether their launches were given up.
usage: [HSSFSST_TEAM_ONLY=1] t16_probe.py lib.so [batch n]"""
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
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    L.hssfsst_plan_fallbacks.argtypes = [ctypes.c_void_p]
    w = np.ascontiguousarray(synth.kaiser_window(128, 0.5))
    plan = ctypes.c_void_p()
    rc = L.hssfsst_plan_create(ctypes.byref(plan), 0, 128, w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 1000.0, 1, 25.0, 200.0, 2)
    assert rc == 0, L.hssfsst_last_error()
    X = torch.from_numpy((synth.pcg_windows(B, n) + float(os.environ.get("T16_OFFSET", "0"))).astype(np.float32)).cuda()
    out = torch.empty((B, n, 44), dtype=torch.float32, device="cuda")
    for rd in range(int(os.environ.get("T16_ROUNDS", "4"))):
        L.hssfsst_plan_set_timing(plan, 0 if os.environ.get("T16_NOTIMING") else 1)
        torch.cuda.synchronize(); import time as _t; _w0 = _t.perf_counter()
        for _ in range(int(os.environ.get("T16_EXECS", "100"))):
            rc = L.hssfsst_exec(plan, ctypes.c_void_p(X.data_ptr()), B, n, 1, ctypes.c_void_p(out.data_ptr()), 1, None)
            assert rc == 0, L.hssfsst_last_error()
        ms = (ctypes.c_float * 2)(); cnt = ctypes.c_int()
        L.hssfsst_plan_timing(plan, ms, ctypes.byref(cnt))
        torch.cuda.synchronize(); _wall = (_t.perf_counter() - _w0) / max(cnt.value, 1) * 1e3
        fb = L.hssfsst_plan_fallbacks(plan)
        if cnt.value == 0:
            torch.cuda.synchronize(); print(f"round {rd}: wall {(_t.perf_counter() - _w0) / int(os.environ.get('T16_EXECS', '100')) * 1e3:.4f} ms/exec (no events)", flush=True); continue
        t = (ms[0] + ms[1]) / cnt.value
        print(f"round {rd}: {B}x{n} {t:.4f} ms/exec (core {ms[0] / cnt.value:.4f})  {B / t / 1e3:.3f} Mwin/s  {(8000 + 352000) * (n / 2000) * B / (t * 1e-3) / 8e12 * 100:.2f}% of 8 TB/s  "
              f"zpath={L.hssfsst_plan_last_exec_fused(plan)} fallbacks={fb}  wall {_wall:.4f} ms/exec", flush=True)
        if hasattr(L, "hssfsst_dev_t16_probe"):
            buf = (ctypes.c_ulonglong * 16)()
            L.hssfsst_dev_t16_probe(buf)
            wv = max(buf[12], 1)
            if os.environ.get("T16_WAITS"):
                if buf[6]:
                    print(f"   resolves {buf[6]}: copy {buf[4] / buf[6] / 100:.2f} us in {buf[7] / buf[6]:.2f} looks, sums {buf[5] / buf[6] / 100:.2f} us each", flush=True)
                print(f"   per wave: blocked {buf[0] / wv / 100:.1f} us in {buf[1] / wv:.1f} waits ({buf[0] / max(buf[1], 1) / 100:.2f} us each), found ready {buf[2] / wv:.1f} times, lifetime {buf[3] / wv / 100:.1f} us", flush=True)
                continue
            names = ["transform", "land", "stats+publish+draw", "emit", "wait(all)", "resolver poll", "resolver compute", "image", "loop top"]
            print("   per wave (cycles): " + "  ".join(f"{nm} {buf[k] / wv:.0f}" for k, nm in enumerate(names)) +
                  f" | lifetime {buf[11] / wv:.0f} resolves/wave {buf[9] / wv:.2f} groups/wave {buf[10] / wv:.1f}", flush=True)


if __name__ == "__main__":
    main()
