#!/usr/bin/env python3
"""Development check of one build of libhssfsst.so: runs the canonical STACK transform on seeded PCG windows for several
(batch, n) shapes, prints time per exec, and saves the outputs so that two runs (e.g. team kernel vs HSSFSST_NO_FUSED=1)
can be compared bit for bit.   usage: team_check.py lib.so out.npz [B:n ...]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import synth  # noqa: E402


def main():
    path, outp = sys.argv[1], sys.argv[2]
    shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[3:]] or [(1024, 2000)]
    L = ctypes.CDLL(path)
    vp, ip, dp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)
    L.hssfsst_plan_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.c_int, dp, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int]
    L.hssfsst_exec.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp]
    L.hssfsst_plan_set_timing.argtypes = [vp, ctypes.c_int]
    L.hssfsst_plan_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ip]
    L.hssfsst_plan_check.argtypes = [vp]
    L.hssfsst_plan_last_exec_fused.argtypes = [vp]
    L.hssfsst_last_error.restype = ctypes.c_char_p
    w = np.ascontiguousarray(synth.kaiser_window(128, 0.5))
    plan = vp()
    rc = L.hssfsst_plan_create(ctypes.byref(plan), 0, 128, w.ctypes.data_as(dp), 1000.0, 1, 25.0, 200.0, 2)
    assert rc == 0, L.hssfsst_last_error()
    res = {}
    for (B, n) in shapes:
        X = torch.from_numpy(synth.pcg_windows(B, n, seed=B + n)).cuda()
        out = torch.full((B, n, 44), float("nan"), dtype=torch.float32, device="cuda")
        steps = int(os.environ.get("TC_STEPS", "30"))
        for it in range(3):
            L.hssfsst_plan_set_timing(plan, 1)
            for _ in range(steps):
                rc = L.hssfsst_exec(plan, vp(X.data_ptr()), B, n, 1, vp(out.data_ptr()), 1, None)
                assert rc == 0, L.hssfsst_last_error()
            ms = (ctypes.c_float * 2)(); cnt = ctypes.c_int()
            L.hssfsst_plan_timing(plan, ms, ctypes.byref(cnt))
        rc = L.hssfsst_plan_check(plan)
        assert rc == 0, L.hssfsst_last_error()
        o = out.cpu().numpy()
        t = (ms[0] + ms[1]) / cnt.value
        print(f"{os.path.basename(path)} B={B} n={n}: {t:.4f} ms/exec ({B / t / 1e3:.3f} Mwin/s, {(8000 * n / 2000 + n * 176) * B / (t * 1e-3) / 8e12 * 100:.2f}% of 8 TB/s) "
              f"fused={L.hssfsst_plan_last_exec_fused(plan)} nan={int(np.isnan(o).sum())} mean={o.mean():.3e} std={o.std():.6f}", flush=True)
        import zlib
        res[f"{B}x{n}_crc"] = np.array([zlib.crc32(o.tobytes())], dtype=np.uint32)       # whole output
        res[f"{B}x{n}_head"] = o[:2]
    np.savez(outp, **res)


if __name__ == "__main__":
    main()
