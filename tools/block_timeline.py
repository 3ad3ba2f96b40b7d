#!/usr/bin/env python3
"""Wave residency timeline of the nwin=128 core (library built with -DHSS_CLOCKPROBE=2): one record per wave.
usage: [NBLOCKS=4096 BY_RANK=1 BY_XCD=1] block_timeline.py lib.so   (NBLOCKS = number of wave records = grid * waves per block)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import synth
from tools.ab_bench import load
L = load(sys.argv[1]); plan = ctypes.c_void_p()
w = np.ascontiguousarray(synth.kaiser_window(128, 0.5))
assert L.hssfsst_plan_create(ctypes.byref(plan), 0, 128, w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 1000.0, 1, 25.0, 200.0, 3) == 0
B = 1024
nblocks = int(os.environ.get("NBLOCKS", "4096"))
X = torch.from_numpy(synth.pcg_windows(B, 2000)).cuda(); out = torch.empty((B, 2000, 44), dtype=torch.float32, device="cuda")
for _ in range(600):
    assert L.hssfsst_exec(plan, ctypes.c_void_p(X.data_ptr()), B, 2000, 1, ctypes.c_void_p(out.data_ptr()), 1, None) == 0
torch.cuda.synchronize()
t = out.view(torch.int32).flatten()[: 2 * nblocks].cpu().numpy().astype(np.int64).reshape(-1, 2) & 0xffffffff
ok = (t[:, 1] >= t[:, 0]) & (t[:, 1] - t[:, 0] < 100000)
t = t[ok]; t0 = t[:, 0].min(); s = (t[:, 0] - t0) / 100.0; e = (t[:, 1] - t0) / 100.0      # microseconds
print(f"{ok.sum()} of {nblocks} wave records usable; kernel span {e.max():.1f} us; wave life median {np.median(e - s):.1f} us "
      f"(p5 {np.percentile(e - s, 5):.1f}, p95 {np.percentile(e - s, 95):.1f})")
edges = np.arange(0, e.max() + 10, 10.0)
for a in edges[:-1]:
    resident = ((s < a + 10) & (e > a)).sum()
    mid = ((s <= a + 5) & (e > a + 5)).sum()
    print(f"t = {a:5.0f}..{a + 10:5.0f} us: waves resident at the midpoint {mid:5d} (capacity 4096), started in the slice {((s >= a) & (s < a + 10)).sum():5d}")
if os.environ.get("BY_XCD"):
    life = (e - s)
    idx = np.nonzero(ok)[0]
    for x in range(8):
        m = (idx % 8) == x
        print(f"blockIdx % 8 == {x}: life mean {life[m].mean():6.1f} us  min {life[m].min():6.1f}  max {life[m].max():6.1f}")
    for c in range(0, 32, 4):
        m = ((idx // 8) % 32) == c
        print(f"(blockIdx / 8) % 32 == {c:2d}: life mean {life[m].mean():6.1f} us")
    order = np.argsort(life)
    print("slowest blocks:", idx[order[-12:]], "fastest:", idx[order[:12]])
if os.environ.get("BY_RANK"):
    life = (e - s); idx = np.nonzero(ok)[0]
    for r in range(0, (nblocks + 255) // 256):
        m = (idx // 256) == r
        print(f"blockIdx / 256 == {r}: start mean {s[m].mean():6.1f}  end mean {e[m].mean():6.1f} (p5 {np.percentile(e[m], 5):6.1f}, p95 {np.percentile(e[m], 95):6.1f}) us")
if os.environ.get("BY_BLOCK"):
    wpb = int(os.environ["BY_BLOCK"]); idx = np.nonzero(ok)[0]
    nb = nblocks // wpb
    bend = np.array([e[(idx // wpb) == b].max() for b in range(nb)]); bmin = np.array([e[(idx // wpb) == b].min() for b in range(nb)])
    print(f"per block of {wpb} waves: last-wave end  mean {bend.mean():.1f}  p5 {np.percentile(bend, 5):.1f}  p95 {np.percentile(bend, 95):.1f}  max {bend.max():.1f} us; "
          f"first-to-last wave end inside a block: mean {np.mean(bend - bmin):.1f} us")
    for x in range(8):
        print(f"  blocks with blockIdx % 8 == {x}: last-wave end mean {bend[np.arange(nb) % 8 == x].mean():.1f} us")
