#!/usr/bin/env python3
"""Ticket-level time budget of the one-CU-per-signal kernel (library built with -DHSS_FUSE_PROBE): shader-clock cycles per wave spent in
transform tickets (A), in z-score tickets (B: waiting for statistics / loads in flight / arithmetic + stores) and in the resolver.
usage: fuse_probe2.py lib.so"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import synth
L = ctypes.CDLL(sys.argv[1])
vp, dp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)
L.hssfsst_plan_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.c_int, dp, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int]
L.hssfsst_exec.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp]
L.hssfsst_plan_set_timing.argtypes = [vp, ctypes.c_int]
L.hssfsst_plan_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
w = np.ascontiguousarray(synth.kaiser_window(128, 0.5))
plan = vp()
assert L.hssfsst_plan_create(ctypes.byref(plan), 0, 128, w.ctypes.data_as(dp), 1000.0, 1, 25.0, 200.0, 2) == 0
X = torch.from_numpy(synth.pcg_windows(1024, 2000)).cuda()
out = torch.empty((1024, 2000, 44), dtype=torch.float32, device="cuda")
def run(k):
    for _ in range(k): L.hssfsst_exec(plan, vp(X.data_ptr()), 1024, 2000, 1, vp(out.data_ptr()), 1, None)
run(300)
h = (ctypes.c_ulonglong * 8)()
L.hssfsst_dev_fuse_probe(h)
L.hssfsst_plan_set_timing(plan, 1)
N = 200
run(N)
ms = (ctypes.c_float * 2)(); cnt = ctypes.c_int()
L.hssfsst_plan_timing(plan, ms, ctypes.byref(cnt))
L.hssfsst_dev_fuse_probe(h)
waves = 256 * 16 * N
print(f"kernel {(ms[0] + ms[1]) / cnt.value:.4f} ms per launch (with probes)")
a_n, a_c, b_n, b_w, b_l, b_s, rs = [h[i] for i in range(7)]
print(f"per wave and launch: {a_n / waves:.1f} A tickets, {a_c / waves:.0f} cycles ({a_c / max(a_n, 1):.0f} per ticket); {b_n / waves:.1f} B tickets: wait for statistics "
      f"{b_w / waves:.0f}, loads in flight {b_l / waves:.0f}, arithmetic + stores issued {b_s / waves:.0f} cycles ({(b_w + b_l + b_s) / max(b_n, 1):.0f} per ticket: "
      f"{b_w / max(b_n, 1):.0f} / {b_l / max(b_n, 1):.0f} / {b_s / max(b_n, 1):.0f}); resolver {rs / waves:.0f}")
