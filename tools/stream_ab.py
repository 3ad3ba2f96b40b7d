#!/usr/bin/env python3
"""A/B of builds of libhssfsst.so on the streaming step (BASELINE config 5: 64 channels x 4 kHz, chunk 128, Kaiser(512)).
usage: stream_ab.py lib_a.so [lib_b.so ...]   -- per build (own process): queued steps per second (3 x 2000) and the host-visible
latency of step_host (pinned in, pinned out, one synchronisation), median of 300."""
import os, subprocess, sys, time
if len(sys.argv) > 2 or (len(sys.argv) == 2 and sys.argv[1] != "--one"):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one", lib], check=False)
    sys.exit(0)
lib = sys.argv[2]
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import _lib, synth
_lib.LIB_PATH = os.path.abspath(lib)
from heart_sounds_segmentation_amd.streaming import StreamingFSST
w = synth.kaiser_window(512, 0.5)
st = StreamingFSST(64, 4000.0, w, truncate_freq=(25, 200), chunk=128, normalize=True)
xh = synth.pcg_windows(64, 128 * 64, fs=4000)
x = torch.from_numpy(xh).cuda()
for i in range(200): st.step(x[:, (i % 64) * 128:(i % 64 + 1) * 128], copy=False)
torch.cuda.synchronize()
rates = []
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(2000): st.step(x[:, (i % 64) * 128:(i % 64 + 1) * 128], copy=False)
    torch.cuda.synchronize()
    rates.append(2000 / (time.perf_counter() - t0))
lat = []
for i in range(350):
    c = np.ascontiguousarray(xh[:, (i % 64) * 128:(i % 64 + 1) * 128])
    t0 = time.perf_counter(); st.step_host(c); lat.append(time.perf_counter() - t0)
lat = np.array(lat[50:]) * 1e6
print(f"{os.path.basename(lib):28s} {np.median(rates) / 1e3:6.2f} k steps/s (min {min(rates) / 1e3:.2f}, max {max(rates) / 1e3:.2f});  host-visible {np.median(lat):6.1f} us median, {np.percentile(lat, 99):6.1f} p99   [{st.last_kernel()}]")
