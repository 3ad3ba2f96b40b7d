#!/usr/bin/env python3
"""Development: share of the groups that leave a wave without their signal's statistics being there (builds with -DHSS_T16_MISSPROBE).
usage: miss_probe.py lib.so [pcg|noise|zeros]"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import synth  # noqa: E402
from tools.canon_check import load  # noqa: E402
L = load(sys.argv[1]); kind = sys.argv[2] if len(sys.argv) > 2 else "pcg"
B, n = 1024, 2000
w = np.ascontiguousarray(synth.kaiser_window(128, 0.5))
plan = ctypes.c_void_p()
assert L.hssfsst_plan_create(ctypes.byref(plan), 0, 128, w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 1000.0, 1, 25.0, 200.0, 2) == 0
xh = {"pcg": lambda: synth.pcg_windows(B, n), "noise": lambda: synth.noise_windows(B, n), "zeros": lambda: np.zeros((B, n), np.float32)}[kind]()
X = torch.from_numpy(xh.astype(np.float32)).cuda(); out = torch.empty((B, n, 44), dtype=torch.float32, device="cuda")
buf = (ctypes.c_ulonglong * 1032)(); L.hssfsst_dev_t16_xcc.argtypes = [ctypes.c_void_p, ctypes.c_int]
def run(k):
    for _ in range(k): L.hssfsst_exec(plan, ctypes.c_void_p(X.data_ptr()), B, n, 1, ctypes.c_void_p(out.data_ptr()), 1, None)
run(300)
for r in range(3):
    assert L.hssfsst_dev_t16_xcc(None, 1) == 0
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(100); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    assert L.hssfsst_dev_t16_xcc(buf, 0) == 0
    a = np.frombuffer(buf, dtype=np.uint64)
    print(f"{os.path.basename(sys.argv[1])} {kind}: {dt * 1e6:.1f} us per exec; {a[1024]} of {a[1025]} leaving groups without statistics in time = {100.0 * a[1024] / max(a[1025], 1):.2f} %")
