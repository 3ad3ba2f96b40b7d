#!/usr/bin/env python3
"""Development: how long waves of the team kernel wait for statistics (builds with -DHSS_T16_BLKPROBE: three counters per wave, one atomic each at the end).
usage: blk_probe.py lib.so [pcg|noise|zeros]"""
import ctypes, os, sys
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
buf = (ctypes.c_uint * (256 * 16 * 8))(); L.hssfsst_dev_t16_blk.argtypes = [ctypes.c_void_p]
def run(k):
    for _ in range(k): L.hssfsst_exec(plan, ctypes.c_void_p(X.data_ptr()), B, n, 1, ctypes.c_void_p(out.data_ptr()), 1, None)
run(300)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(3):
    K = 200
    e0.record(); run(K); e1.record(); e1.synchronize()
    assert L.hssfsst_dev_t16_blk(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint32).reshape(4096, 8).astype(np.float64)
    miss, blocked, fin, nfin = a[:, 0], a[:, 2] / 100.0, a[:, 3] / 100.0, a[:, 4]
    print(f"{os.path.basename(sys.argv[1])} {kind}: {e0.elapsed_time(e1) / K * 1e3:.1f} us per exec; last launch: misses {miss.sum() / 128000 * 100:.1f} % of the groups, "
          f"{blocked.sum() / max(miss.sum(), 1):.2f} us waited per miss = {blocked.mean():.2f} us per wave (p90 {np.percentile(blocked, 90):.2f}, max {blocked.max():.2f}); "
          f"resolvers take {fin.sum() / max(nfin.sum(), 1):.2f} us per signal = {fin.mean():.2f} us per wave; shader clock {100.0 * a[:, 5].sum() / max(a[:, 6].sum(), 1):.0f} MHz (cycle counter / 100 MHz clock over the waves' lifetimes: mean {a[:, 6].mean() / 100:.1f} us)")
    ent, end = a[:, 1], a[:, 7]
    t0 = ent.min()
    ent, end = (ent - t0) / 100.0, (end - t0) / 100.0        # us since the first wave's first instruction
    loop0 = end - a[:, 6] / 100.0                             # the wave's first draw (behind table staging and the block's barrier)
    team_end = end.reshape(16, 256).max(axis=1)               # 16 teams x (16 CUs x 16 waves)
    cu_end = end.reshape(256, 16).max(axis=1)
    print(f"   timeline (us since the first wave entered): waves enter {ent.mean():.1f} +- {ent.std():.1f} (last {ent.max():.1f}); first draw {loop0.mean():.1f} (last {loop0.max():.1f}); "
          f"waves end {end.mean():.1f} +- {end.std():.1f}, first {end.min():.1f}, last {end.max():.1f}; teams end {team_end.min():.1f} .. {team_end.max():.1f} (mean {team_end.mean():.1f}); "
          f"CUs end mean {cu_end.mean():.1f}")
    print("   teams end (us), in arrival order:", " ".join(f"{v:.1f}" for v in team_end), "| CUs of team 0 end:", " ".join(f"{v:.1f}" for v in cu_end[:16]))
