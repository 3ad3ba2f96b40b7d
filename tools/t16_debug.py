#!/usr/bin/env python3
"""Development: one exec of the 16-wave team kernel built with -DHSS_T16_DEBUG; prints the status word of a wait that gave up."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import synth
from tools.canon_check import load
L = load(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
w = np.ascontiguousarray(synth.kaiser_window(128, 0.5))
plan = ctypes.c_void_p()
assert L.hssfsst_plan_create(ctypes.byref(plan), 0, 128, w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 1000.0, 1, 25.0, 200.0, 2) == 0
X = torch.from_numpy(synth.pcg_windows(B, 2000)).cuda()
out = torch.empty((B, 2000, 44), dtype=torch.float32, device="cuda")
for it in range(3):
    rc = L.hssfsst_exec(plan, ctypes.c_void_p(X.data_ptr()), B, 2000, 1, ctypes.c_void_p(out.data_ptr()), 1, None)
    print("exec rc", rc, L.hssfsst_last_error() if rc else "")
    rc = L.hssfsst_plan_check(plan)
    print("check rc", rc, L.hssfsst_last_error() if rc else "")
    if hasattr(L, "hssfsst_dev_t16_dbg"):
        buf = (ctypes.c_uint * 128)()
        L.hssfsst_dev_t16_dbg(buf)
        print("dbg count", buf[0])
        for k in range(min(buf[0], 15)):
            a, b_, c, d = buf[4 * k + 4: 4 * k + 8]
            pair = b_ & 0xffff
            print(f"  team {a >> 24} member {(a >> 16) & 0xff} ko {a & 0xffff}: first missing pair {pair} (group {pair * 2 // 6} word {pair * 2 % 6}) need {b_ >> 16:06b} tag seen {c:#x} want {d:#x}")
        print("waiters recorded", buf[1])
        for k in range(min(buf[1], 20)):
            a, b_ = buf[64 + 2 * k], buf[65 + 2 * k]
            print(f"  team {a >> 24} member {(a >> 16) & 0xff} wave {(a >> 8) & 0xff} waits for ko {a & 0xff} (held group {b_ >> 16}); current: ko {(b_ >> 8) & 0xff} group {b_ & 0xff}")
