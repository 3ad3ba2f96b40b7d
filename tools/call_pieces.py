#!/usr/bin/env python3
"""Development: the Python-side pieces of one drop-in FSST.__call__ (microseconds each)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import FSST, synth, _lib
tf = FSST(1000, synth.kaiser_window(128, 0.5), truncate_freq=(25, 200), stack=True)
x = torch.from_numpy(synth.pcg_windows(1, 2000)[0]).reshape(2000, 1)
tf(x)
def t(f, k=20000):
    for _ in range(200): f()
    t0 = time.perf_counter()
    for _ in range(k): f()
    return (time.perf_counter() - t0) / k * 1e6
print(f"guard_fork            {t(_lib.guard_fork):.2f}")
print(f"cuda.is_available     {t(torch.cuda.is_available):.2f}")
print(f"cuda.current_device   {t(torch.cuda.current_device):.2f}")
print(f"_device_index(x)      {t(lambda: tf._device_index(x)):.2f}")
print(f"_mode                 {t(tf._mode):.2f}")
print(f"_plan(0)              {t(lambda: tf._plan(0)):.2f}")
print(f"_lib.lib()            {t(_lib.lib):.2f}")
print(f"os.getpid             {t(os.getpid):.2f}")
print(f"__call__              {t(lambda: tf(x), 3000):.2f}")
