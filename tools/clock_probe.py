#!/usr/bin/env python3
"""Sustained shader clock inside the nwin=128 core (library built with -DHSS_CLOCKPROBE).  usage: clock_probe.py lib.so"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import synth
from tools.ab_bench import load
L = load(sys.argv[1]); plan = ctypes.c_void_p()
w = np.ascontiguousarray(synth.kaiser_window(128, 0.5))
assert L.hssfsst_plan_create(ctypes.byref(plan), 0, 128, w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 1000.0, 1, 25.0, 200.0, 3) == 0
B = 1024
X = torch.from_numpy(synth.pcg_windows(B, 2000)).cuda(); out = torch.empty((B, 2000, 44), dtype=torch.float32, device="cuda")
for steps in (5, 50, 500, 2000):
    for _ in range(steps):
        assert L.hssfsst_exec(plan, ctypes.c_void_p(X.data_ptr()), B, 2000, 1, ctypes.c_void_p(out.data_ptr()), 1, None) == 0
    torch.cuda.synchronize()
    c, r = out[0, 0, 0].item(), out[0, 0, 1].item()
    print(f"after {steps:5d} more back-to-back launches: wave lived {c:.0f} shader ticks / {r:.0f} ticks @100 MHz  =>  {100.0 * c / r:.0f} MHz")
