#!/usr/bin/env python3
"""Development: timeline of team 0 of the team kernel's last launch (builds with -DHSS_T16_TLPROBE).  usage: timeline.py lib.so [pcg|noise|zeros]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import synth  # noqa: E402
from tools.canon_check import load  # noqa: E402
L = load(sys.argv[1]); kind = sys.argv[2] if len(sys.argv) > 2 else "pcg"
B, n, CAP = 1024, 2000, 512
w = np.ascontiguousarray(synth.kaiser_window(128, 0.5))
plan = ctypes.c_void_p()
assert L.hssfsst_plan_create(ctypes.byref(plan), 0, 128, w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 1000.0, 1, 25.0, 200.0, 2) == 0
xh = {"pcg": lambda: synth.pcg_windows(B, n), "noise": lambda: synth.noise_windows(B, n), "zeros": lambda: np.zeros((B, n), np.float32)}[kind]()
X = torch.from_numpy(xh.astype(np.float32)).cuda(); out = torch.empty((B, n, 44), dtype=torch.float32, device="cuda")
N = 16 * 16 * CAP * 2
buf = (ctypes.c_uint * N)(); L.hssfsst_dev_t16_tl.argtypes = [ctypes.c_void_p, ctypes.c_int]
def run(k):
    for _ in range(k): L.hssfsst_exec(plan, ctypes.c_void_p(X.data_ptr()), B, n, 1, ctypes.c_void_p(out.data_ptr()), 1, None)
run(300)
assert L.hssfsst_dev_t16_tl(None, 1) == 0
run(30)
assert L.hssfsst_dev_t16_tl(buf, 0) == 0
a = np.frombuffer(buf, dtype=np.uint32).reshape(16, 16, CAP, 2)
ev = a[..., 0] >> 28; ko = (a[..., 0] >> 8) & 0xfffff; g = a[..., 0] & 0xff; t = a[..., 1].astype(np.int64)
valid = ev > 0
t0 = t[valid].min()
tt = (t - t0) / 100.0                                   # us
print(f"{os.path.basename(sys.argv[1])} {kind}: team 0, {valid.sum()} events, span {tt[valid].max():.1f} us")
nsig = int(ko[valid].max()) + 1
pub_last = np.full(nsig, -1.0); pub_first = np.full(nsig, 1e9); fin_start = np.full(nsig, np.nan); fin_done = np.full(nsig, np.nan)
pub_cu_last = np.full((nsig, 16), -1.0)
for c in range(16):
    for wv in range(16):
        m = valid[c, wv]
        for e, k, tm in zip(ev[c, wv][m], ko[c, wv][m], tt[c, wv][m]):
            if e == 1:
                pub_last[k] = max(pub_last[k], tm); pub_first[k] = min(pub_first[k], tm); pub_cu_last[k, c] = max(pub_cu_last[k, c], tm)
            elif e == 2: fin_start[k] = tm
            elif e == 3: fin_done[k] = tm
sel = slice(4, nsig - 4)
print(f"signals {nsig}; per signal (us): first->last publish {np.nanmean((pub_last - pub_first)[sel]):.2f} (p90 {np.nanpercentile((pub_last - pub_first)[sel], 90):.2f}), "
      f"finisher start - last publish {np.nanmean((fin_start - pub_last)[sel]):.2f}, finisher done - last publish {np.nanmean((fin_done - pub_last)[sel]):.2f} (p90 {np.nanpercentile((fin_done - pub_last)[sel], 90):.2f})")
d = np.diff(fin_done[sel]); print(f"signal completion spacing: mean {np.nanmean(d):.2f} us, std {np.nanstd(d):.2f}, p10 {np.nanpercentile(d, 10):.2f} p90 {np.nanpercentile(d, 90):.2f}")
cu_skew = pub_cu_last[sel].max(1) - pub_cu_last[sel].min(1)
print(f"CU skew per signal (latest CU's last publish - earliest CU's): mean {cu_skew.mean():.2f} us, p90 {np.percentile(cu_skew, 90):.2f}; which CU is last most often: {np.bincount(pub_cu_last[sel].argmax(1), minlength=16)}")
# takes
hits = miss = 0; lead = []; blocked = []; hit_margin = []
for c in range(16):
    for wv in range(16):
        m = valid[c, wv]; E = ev[c, wv][m]; K = ko[c, wv][m]; T = tt[c, wv][m]
        for i in range(len(E)):
            if E[i] == 4: hits += 1; hit_margin.append(T[i] - fin_done[K[i]])
            if E[i] == 5:
                miss += 1; lead.append(T[i] - pub_last[K[i]])
                if i + 1 < len(E) and E[i + 1] == 6: blocked.append(T[i + 1] - T[i])
print(f"takes: {hits} hits, {miss} misses ({100.0 * miss / max(hits + miss, 1):.1f} %); a hit comes {np.nanmean(hit_margin):.2f} us after the finisher was done (p10 {np.nanpercentile(hit_margin, 10):.2f}); "
      f"a miss comes {np.mean(lead) if lead else 0:.2f} us after the signal's last publish (negative: before) and blocks {np.mean(blocked) if blocked else 0:.2f} us (p90 {np.percentile(blocked, 90) if blocked else 0:.2f})")
# step durations
steps = []
for c in range(16):
    for wv in range(16):
        m = valid[c, wv]; E = ev[c, wv][m]; T = tt[c, wv][m]
        s7 = T[E == 7]; steps.extend(np.diff(s7))
steps = np.asarray(steps); print(f"step time per wave: mean {steps.mean():.2f} us, std {steps.std():.2f}, p50 {np.percentile(steps, 50):.2f} p90 {np.percentile(steps, 90):.2f} p99 {np.percentile(steps, 99):.2f} max {steps.max():.2f}")
np.save("gpurun_out/timeline_%s.npy" % kind, a)
