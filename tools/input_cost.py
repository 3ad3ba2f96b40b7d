import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from heart_sounds_segmentation_amd import FSST, synth
tf = FSST(1000, synth.kaiser_window(128, 0.5), truncate_freq=(25, 200), stack=True)
out = torch.empty((1024, 2000, 44), dtype=torch.float32, device="cuda")
t = np.arange(2000) / 1000.0
inputs = {
    "pcg": synth.pcg_windows(1024, 2000),
    "noise": synth.noise_windows(1024, 2000),
    "zeros (no cell moves)": np.zeros((1024, 2000), np.float32),
    "tone 125 Hz on-bin": np.tile(np.cos(2 * np.pi * 125.0 * t).astype(np.float32), (1024, 1)),
    "ramp+dc": np.tile((1.0 + t).astype(np.float32), (1024, 1)),
    "pcg + 3.0": (synth.pcg_windows(1024, 2000) + 3.0).astype(np.float32),
    "pcg + 100": (synth.pcg_windows(1024, 2000) + 100.0).astype(np.float32),
    "constant 5": np.full((1024, 2000), 5.0, np.float32),
}
X0 = torch.from_numpy(inputs["pcg"]).cuda()
for _ in range(400): tf.batch(X0, out=out)
for name, x in inputs.items():
    X = torch.from_numpy(x).cuda()
    for _ in range(50): tf.batch(X, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500): tf.batch(X, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 500
    print(f"{name:24s} {dt*1e3:.4f} ms per 1024 windows")
