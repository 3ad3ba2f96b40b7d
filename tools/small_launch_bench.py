#!/usr/bin/env python3
"""Latency of small launches (device-resident, STACK, canonical plan).  usage: small_launch_bench.py lib.so [lib2.so ...]"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import synth
from tools.ab_bench import load
w = np.ascontiguousarray(synth.kaiser_window(128, 0.5))
for path in sys.argv[1:]:
    L = load(path); plan = ctypes.c_void_p()
    assert L.hssfsst_plan_create(ctypes.byref(plan), 0, 128, w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 1000.0, 1, 25.0, 200.0, 2) == 0
    for B in (1, 4, 33, 128):
        X = torch.from_numpy(synth.pcg_windows(B, 2000)).cuda(); out = torch.empty((B, 2000, 44), dtype=torch.float32, device="cuda")
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(200):
                assert L.hssfsst_exec(plan, ctypes.c_void_p(X.data_ptr()), B, 2000, 1, ctypes.c_void_p(out.data_ptr()), 1, None) == 0
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        L.hssfsst_plan_set_timing(plan, 1)
        for _ in range(50):
            assert L.hssfsst_exec(plan, ctypes.c_void_p(X.data_ptr()), B, 2000, 1, ctypes.c_void_p(out.data_ptr()), 1, None) == 0
        ms = (ctypes.c_float * 2)(); cnt = ctypes.c_int()
        L.hssfsst_plan_timing(plan, ms, ctypes.byref(cnt)); L.hssfsst_plan_set_timing(plan, 0)
        print(f"{os.path.basename(path):14s} batch {B:4d}: {dt * 1e6:8.1f} us per exec (back-to-back, device buffers); "
              f"events: core {ms[0] / cnt.value * 1e3:6.1f} us, rest {ms[1] / cnt.value * 1e3:6.1f} us")
    # host buffers in and out (the drop-in __call__ path): one window
    xh = np.ascontiguousarray(synth.pcg_windows(1, 2000)); oh = np.empty((1, 2000, 44), np.float32)
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(200):
            assert L.hssfsst_exec(plan, xh.ctypes.data_as(ctypes.c_void_p), 1, 2000, 0, oh.ctypes.data_as(ctypes.c_void_p), 0, None) == 0
        dt = (time.perf_counter() - t0) / 200
    print(f"{os.path.basename(path):14s} one window, host buffers in/out: {dt * 1e6:8.1f} us per exec")
