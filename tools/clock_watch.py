#!/usr/bin/env python3
"""Development: shader clock and socket power (rocm-smi) while a build runs the C2 workload back to back.
usage: clock_watch.py lib.so [pcg|noise|zeros] [seconds]"""
import ctypes, os, subprocess, sys, threading, time, re
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import synth  # noqa: E402
from tools.canon_check import load  # noqa: E402
L = load(sys.argv[1]); kind = sys.argv[2] if len(sys.argv) > 2 else "pcg"; secs = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
B, n = 1024, 2000
w = np.ascontiguousarray(synth.kaiser_window(128, 0.5))
plan = ctypes.c_void_p()
assert L.hssfsst_plan_create(ctypes.byref(plan), 0, 128, w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 1000.0, 1, 25.0, 200.0, 2) == 0
xh = {"pcg": lambda: synth.pcg_windows(B, n), "noise": lambda: synth.noise_windows(B, n), "zeros": lambda: np.zeros((B, n), np.float32)}[kind]()
X = torch.from_numpy(xh.astype(np.float32)).cuda(); out = torch.empty((B, n, 44), dtype=torch.float32, device="cuda")
samples, stop = [], False
def watch():
    while not stop:
        try:
            r = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = re.search(r"sclk clock level.*\((\d+)Mhz\)", r); pw = re.search(r"Power \(W\):\s*([\d.]+)", r) or re.search(r"Socket Power.*?:\s*([\d.]+)", r)
            samples.append((int(sclk.group(1)) if sclk else -1, float(pw.group(1)) if pw else -1.0))
        except Exception as e:
            samples.append((-2, -2.0))
        time.sleep(0.05)
th = threading.Thread(target=watch); th.start()
t0 = time.perf_counter(); k = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.perf_counter() - t0 < secs:
    for _ in range(200): L.hssfsst_exec(plan, ctypes.c_void_p(X.data_ptr()), B, n, 1, ctypes.c_void_p(out.data_ptr()), 1, None)
    k += 200
    torch.cuda.synchronize()
e1.record(); e1.synchronize()
stop = True; th.join()
s = np.asarray(samples[2:]) if len(samples) > 4 else np.asarray(samples)
print(f"{os.path.basename(sys.argv[1])} {kind}: {e0.elapsed_time(e1) / k * 1e3:.1f} us per exec over {k} execs; rocm-smi x{len(s)}: sclk median {np.median(s[:, 0]):.0f} MHz (min {s[:, 0].min():.0f}), power median {np.median(s[:, 1]):.0f} W (max {s[:, 1].max():.0f})")
