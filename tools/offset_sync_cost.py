import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from heart_sounds_segmentation_amd import FSST, synth
tf = FSST(1000, synth.kaiser_window(128, 0.5), truncate_freq=(25, 200), stack=True)
out = torch.empty((1024, 2000, 44), dtype=torch.float32, device="cuda")
base = synth.pcg_windows(1024, 2000)
for name, x in (("pcg", base), ("pcg + 3.0", base + 3.0), ("1 + t", np.tile(1.0 + np.arange(2000) / 1000.0, (1024, 1)))):
    X = torch.from_numpy(x.astype(np.float32)).cuda()
    for _ in range(10): tf.batch(X, out=out); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): tf.batch(X, out=out); torch.cuda.synchronize()
    print(f"{name:10s} {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per exec, synchronised after every exec; fallbacks {tf.fallbacks()} kernel {tf.last_kernel()}")
