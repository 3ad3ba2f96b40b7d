#!/usr/bin/env python3
"""Development probe of the fused z-score kernel (build with -DHSS_FUSEPROBE): per-wave shader-clock ticks spent in
stage / compute+publish / wait-for-statistics / normalise.  usage (GPU box): HIPCC_FLAGS=-DHSS_FUSEPROBE python tools/fuse_probe.py"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heart_sounds_segmentation_amd import _lib
so = "/tmp/libhssfsst_probe.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DHSS_FUSEPROBE",
                "-o", so, _lib.SRC], check=True)
_lib.LIB_PATH = so
import torch
from heart_sounds_segmentation_amd import FSST, synth
B, n = 1024, 2000
X = torch.from_numpy(synth.pcg_windows(B, n, seed=5)).cuda()
tf = FSST(1000, synth.kaiser_window(128, 0.5), truncate_freq=(25, 200), stack=True)
out = torch.empty((B, n, 44), dtype=torch.float32, device="cuda")
for _ in range(50):
    tf.batch(X, out=out)
torch.cuda.synchronize()
plan = tf._plan(0)
# the probe writes into the plan's partials buffer: fetch its pointer through a second exec's side effect is not exposed,
# so read it via a tiny helper export
L = _lib.lib()
L.hssfsst_debug_partials.restype = ctypes.c_void_p
L.hssfsst_debug_partials.argtypes = [ctypes.c_void_p]
ptr = L.hssfsst_debug_partials(plan.handle)
buf = torch.empty(65536 + 8 * 40 * 4, dtype=torch.float32, device="cuda")
import ctypes as C
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy(C.c_void_p(buf.data_ptr()), C.c_void_p(ptr), C.c_size_t(buf.numel() * 4), 3)
full = buf.cpu().numpy()
a = full[:256 * 16 * 8].reshape(256, 16, 8)[..., :6]
tl = full[65536:].reshape(8, 40, 4)
t0 = tl[:, 0, 0].min()
print("team 0 timeline (us since first publish; 100 MHz ticks / 100): per iteration: publish time of each CU | resolve(start, done, polls) of CU 0 and CU 7")
for k in range(0, 34):
    pub = (tl[:, k, 0] - t0) / 100.0
    r0 = tl[0, k]; r7 = tl[7, k]
    print(f"  k={k:2d} publish min {pub.min():7.2f} max {pub.max():7.2f} | CU0 resolve {(r0[1]-t0)/100:7.2f} -> {(r0[2]-t0)/100:7.2f} polls {int(r0[3]):3d} | CU7 {(r7[1]-t0)/100:7.2f} -> {(r7[2]-t0)/100:7.2f} polls {int(r7[3]):3d}")
names = ["stage", "compute+stats", "B: wait stats", "B: normalise", "R: resolve", "loop top"]
tot = a.sum(-1)
print("ticks per wave (mean over waves), fraction of the wave's total:")
for i, nm in enumerate(names):
    print(f"  {nm:14s} mean {a[..., i].mean():10.0f}  max {a[..., i].max():10.0f}  {100 * a[..., i].sum() / tot.sum():5.1f} %")
print("total mean", tot.mean(), "max", tot.max(), "-> at 2.4 GHz", tot.max() / 2.4e3, "us")
