#!/usr/bin/env python3
"""Development: time of one build on offset inputs (usage: offset_probe.py lib.so)."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import synth
from tools.canon_check import load
L = load(sys.argv[1])
w = np.ascontiguousarray(synth.kaiser_window(128, 0.5))
plan = ctypes.c_void_p()
assert L.hssfsst_plan_create(ctypes.byref(plan), 0, 128, w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 1000.0, 1, 25.0, 200.0, 2) == 0
out = torch.empty((1024, 2000, 44), dtype=torch.float32, device="cuda")
base = synth.pcg_windows(1024, 2000)
for name, x in (("pcg", base), ("pcg + 3", base + 3.0), ("pcg + 30", base + 30.0), ("pcg + 100", base + 100.0), ("noise + 100", synth.noise_windows(1024, 2000) + 100.0)):
    X = torch.from_numpy(x.astype(np.float32)).cuda()
    for _ in range(20): L.hssfsst_exec(plan, ctypes.c_void_p(X.data_ptr()), 1024, 2000, 1, ctypes.c_void_p(out.data_ptr()), 1, None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): L.hssfsst_exec(plan, ctypes.c_void_p(X.data_ptr()), 1024, 2000, 1, ctypes.c_void_p(out.data_ptr()), 1, None)
    torch.cuda.synchronize(); print(f"{name:12s} {(time.perf_counter() - t0) * 10:.4f} ms", flush=True)
