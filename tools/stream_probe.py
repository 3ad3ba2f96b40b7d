#!/usr/bin/env python3
"""Where a streaming step's one launch spends its time (library built with -DHSS_STREAM_PROBE).  usage: stream_probe.py lib.so
Prints, in microseconds from a block's own start (mean over the waves that get there): prologue done, main loop done, arrival known,
moments merged, normalised; and inside a group: tile staged, passes done."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import _lib, synth
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from heart_sounds_segmentation_amd.streaming import StreamingFSST
L = _lib.lib()
w = synth.kaiser_window(512, 0.5)
st = StreamingFSST(64, 4000.0, w, truncate_freq=(25, 200), chunk=128, normalize=True)
x = torch.from_numpy(synth.pcg_windows(64, 128 * 64, fs=4000)).cuda()
for i in range(20):
    st.step(x[:, (i % 64) * 128:(i % 64 + 1) * 128], copy=False)
torch.cuda.synchronize()
NW = 2048
buf = (ctypes.c_ulonglong * (8 * NW))()
L.hssfsst_dev_stream_probe.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
assert L.hssfsst_dev_stream_probe(buf, NW) == 0
acc = np.zeros((8, 2)); 
for rep in range(50):
    st.step(x[:, (rep % 64) * 128:(rep % 64 + 1) * 128], copy=False)
    assert L.hssfsst_dev_stream_probe(buf, NW) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(NW, 8).astype(np.float64)
    for k in range(8):
        nz = a[:, k][a[:, k] > 0]
        acc[k, 0] += nz.sum(); acc[k, 1] += nz.size
names = ["prologue", "main loop", "arrival", "merged", "normalised", "tile staged (last group)", "passes done (last group)", "end"]
for k, n in enumerate(names):
    print(f"{n:28s} {acc[k, 0] / 100.0 / max(1, acc[k, 1]):8.2f} us  (mean over {int(acc[k, 1])} waves)")
print(st.last_kernel())
t0 = time.perf_counter()
for i in range(2000):
    st.step(x[:, (i % 64) * 128:(i % 64 + 1) * 128], copy=False)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 2000 * 1e6:.2f} us per step (probes in)")
