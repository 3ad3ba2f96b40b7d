#!/usr/bin/env python3
"""Robust-column flip census of several builds of libhssfsst.so (raw mode, all 65 rows) against the fp64 oracle.
usage: flip_census.py lib_a.so lib_b.so ...    env: FC_SIGNALS (default 512)
A column is 'robust' when the oracle's nearest rounding boundary is >= parity.FRAG_EPS bins away; a robust column whose
error exceeds parity.TOL means the fp32 shift estimate was off by more than FRAG_EPS for some cell."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from heart_sounds_segmentation_amd import synth
from tests import parity
from scipy.signal import get_window
from tools.ab_bench import load

def run(path, X, w):
    L = load(path); plan = ctypes.c_void_p()
    wd = np.ascontiguousarray(w, dtype=np.float64)
    rc = L.hssfsst_plan_create(ctypes.byref(plan), 0, len(wd), wd.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 1000.0, 0, 0.0, 0.0, 0)
    assert rc == 0, L.hssfsst_last_error()
    B, n = X.shape
    xd = torch.from_numpy(X).cuda(); out = torch.empty((B, 65, n, 2), dtype=torch.float32, device="cuda")
    rc = L.hssfsst_exec(plan, ctypes.c_void_p(xd.data_ptr()), B, n, 1, ctypes.c_void_p(out.data_ptr()), 1, None)
    assert rc == 0, L.hssfsst_last_error()
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    return o[..., 0] + 1j * o[..., 1]

B = int(os.environ.get("FC_SIGNALS", "512"))
for wname, w in (("hann128", get_window("hann", 128, fftbins=False)), ("kaiser128", synth.kaiser_window(128, 0.5)),
                 ("blackman128", get_window("blackman", 128, fftbins=False))):
    X = synth.noise_windows(B, 2000, seed=77)
    ref, hd = oracle.features(X, 1000, w, None, "raw", nthreads=os.cpu_count(), return_halfdist=True)
    for path in sys.argv[1:]:
        got = run(path, X, w)
        scale = np.abs(ref).max(axis=(1, 2), keepdims=True)
        err = (np.abs(got - ref) / scale).max(axis=1)            # [B][n] per column
        robust = hd >= parity.FRAG_EPS
        e = err[robust]
        print(f"{wname:12s} {os.path.basename(path):14s} robust cols {e.size}  >TOL {int((e > parity.TOL).sum())}  >1e-5 {int((e > 1e-5).sum())}  "
              f">1e-6 {int((e > 1e-6).sum())}  max {e.max():.3e}  fragile {int((~robust).sum())} flipped {int((err[~robust] > parity.TOL).sum())}", flush=True)
