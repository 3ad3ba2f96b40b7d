#!/usr/bin/env python3
"""Within-process interleaved A/B of several builds of libhssfsst.so on the C2 workload.
usage: [AB_NWIN=256 AB_FS=1000] ab_bench.py lib_a.so lib_b.so ...   (prints per-build core/normalize kernel ms, median of rounds)"""
import ctypes
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import synth  # noqa: E402


def load(path):
    L = ctypes.CDLL(path)
    vp, ip, dp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)
    L.hssfsst_plan_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.c_int, dp, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int]
    L.hssfsst_exec.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp]
    L.hssfsst_plan_set_timing.argtypes = [vp, ctypes.c_int]
    L.hssfsst_plan_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ip]
    L.hssfsst_last_error.restype = ctypes.c_char_p
    return L


def main():
    libs = sys.argv[1:]
    B = int(os.environ.get("AB_BATCH", "1024"))
    rounds = int(os.environ.get("AB_ROUNDS", "5"))
    steps = int(os.environ.get("AB_STEPS", "10"))
    nwin = int(os.environ.get("AB_NWIN", "128"))
    fs = float(os.environ.get("AB_FS", "1000"))
    w = np.ascontiguousarray(synth.kaiser_window(nwin, 0.5))
    if os.environ.get("AB_WINDOW", "kaiser") == "hann":
        w = np.ascontiguousarray(np.hanning(nwin).astype(np.float64))
    kind = os.environ.get("AB_INPUT", "pcg")                       # pcg | noise | tone (on-bin: the tie path's worst case)
    if kind == "tone":
        xh = np.tile(np.cos(2 * np.pi * (16 * fs / nwin) * np.arange(2000) / fs).astype(np.float32), (B, 1))
    elif kind == "noise":
        xh = synth.noise_windows(B, 2000)
    elif kind == "zeros":
        xh = np.zeros((B, 2000), dtype=np.float32)
    else:
        xh = synth.pcg_windows(B, 2000)
    X = torch.from_numpy(xh).cuda()
    outs, plans, Ls = [], [], []
    for path in libs:
        L = load(path)
        plan = ctypes.c_void_p()
        rc = L.hssfsst_plan_create(ctypes.byref(plan), 0, nwin, w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), fs, 1, 25.0, 200.0, 2)
        assert rc == 0, L.hssfsst_last_error()
        ip = ctypes.POINTER(ctypes.c_int)
        klo, K = ctypes.c_int(), ctypes.c_int()
        L.hssfsst_band.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ip, ip]
        L.hssfsst_band(nwin, fs, 25.0, 200.0, ctypes.byref(klo), ctypes.byref(K))
        out = torch.empty((B, 2000, 2 * K.value), dtype=torch.float32, device="cuda")
        Ls.append(L); plans.append(plan); outs.append(out)
    res = {p: [] for p in libs}
    for rd in range(rounds + 1):
        for i, path in enumerate(libs):
            L, plan, out = Ls[i], plans[i], outs[i]
            L.hssfsst_plan_set_timing(plan, 1)
            for _ in range(steps):
                rc = L.hssfsst_exec(plan, ctypes.c_void_p(X.data_ptr()), B, 2000, 1, ctypes.c_void_p(out.data_ptr()), 1, None)
                assert rc == 0, L.hssfsst_last_error()
            ms = (ctypes.c_float * 2)(); cnt = ctypes.c_int()
            L.hssfsst_plan_timing(plan, ms, ctypes.byref(cnt))
            if rd > 0:
                res[path].append((ms[0] / cnt.value, ms[1] / cnt.value))
    ref = outs[0]
    for i, path in enumerate(libs):
        a = np.asarray(res[path])
        diff = (outs[i] - ref).abs().max().item()
        core, norm = np.median(a[:, 0]), np.median(a[:, 1])
        print(f"{os.path.basename(path):28s} core {core:8.4f} ms (min {a[:,0].min():.4f})  norm {norm:7.4f} ms  "
              f"=> {B / ((core + norm) * 1e-3) / 1e6:6.3f} Mwin/s  core-roofline {(8000 + 2000 * 4 * outs[i].shape[2]) * B / (core * 1e-3) / 8e12 * 100:5.2f}%  maxdiff_vs_first {diff:.2e}")


if __name__ == "__main__":
    main()
