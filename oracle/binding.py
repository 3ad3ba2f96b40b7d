"""ctypes binding of oracle/fsst_oracle.c (test infrastructure only -- see package docstring)."""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib_path() -> str:
    return os.path.join(_HERE, "libhss_oracle.so")


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (oracle/Makefile).  Building the checker is not using it."""
    src = os.path.join(_HERE, "fsst_oracle.c")
    out = lib_path()
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "libhss_oracle.so"], check=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return out


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(lib_path()):
            build()
        lib = ctypes.CDLL(lib_path())
        dp = ctypes.POINTER(ctypes.c_double)
        fp = ctypes.POINTER(ctypes.c_float)
        lib.hss_oracle_dtwin.argtypes = [dp, ctypes.c_int, ctypes.c_double, dp]
        lib.hss_oracle_dtwin.restype = ctypes.c_int
        lib.hss_oracle_fsst.argtypes = [dp, ctypes.c_int, ctypes.c_double, dp, ctypes.c_int,
                                        dp, dp, dp, dp, dp]
        lib.hss_oracle_fsst.restype = ctypes.c_int
        lib.hss_oracle_band.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                        ctypes.c_double, ctypes.POINTER(ctypes.c_int)]
        lib.hss_oracle_band.restype = ctypes.c_int
        lib.hss_oracle_features.argtypes = [fp, ctypes.c_int64, ctypes.c_int, ctypes.c_double, dp,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                            ctypes.c_double, ctypes.c_int, fp, dp, ctypes.c_int]
        lib.hss_oracle_features.restype = ctypes.c_int
        lib.hss_oracle_update_mean.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int64]
        lib.hss_oracle_update_mean.restype = ctypes.c_double
        lib.hss_oracle_update_variance.argtypes = [ctypes.c_double, ctypes.c_double,
                                                   ctypes.c_double, ctypes.c_int64]
        lib.hss_oracle_update_variance.restype = ctypes.c_double
        lib.hss_oracle_max_threads.restype = ctypes.c_int
        _LIB = lib
    return _LIB


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def dtwin(window, fs: float) -> np.ndarray:
    w = np.ascontiguousarray(window, dtype=np.float64).ravel()
    out = np.empty_like(w)
    rc = _lib().hss_oracle_dtwin(_dptr(w), w.size, float(fs), _dptr(out))
    if rc != 0:
        raise RuntimeError(f"hss_oracle_dtwin failed: {rc}")
    return out


def fsst(x, fs: float, window, return_halfdist: bool = False):
    """Same signature/returns as the reference's ``ssq.fsst`` (synchrosqueeze.py:48):
    ``s (nf, nt) complex128, f (nf,), t (nt,)``.  Accepts (n,), (n,1) or (1,n) real input."""
    xd = np.ascontiguousarray(np.asarray(x, dtype=np.float64).ravel())
    w = np.ascontiguousarray(window, dtype=np.float64).ravel()
    n, N = xd.size, w.size
    nf = N // 2 + 1
    sre = np.empty((nf, n), dtype=np.float64)
    sim = np.empty((nf, n), dtype=np.float64)
    f = np.empty(nf, dtype=np.float64)
    t = np.empty(n, dtype=np.float64)
    hd = np.empty(n, dtype=np.float64) if return_halfdist else None
    rc = _lib().hss_oracle_fsst(_dptr(xd), n, float(fs), _dptr(w), N, _dptr(sre), _dptr(sim),
                                _dptr(f), _dptr(t), _dptr(hd) if hd is not None else None)
    if rc < 0:
        raise RuntimeError(f"hss_oracle_fsst failed: {rc}")
    s = sre + 1j * sim
    if return_halfdist:
        return s, f, t, hd
    return s, f, t


def band(N: int, fs: float, f_lo: float, f_hi: float) -> Tuple[int, int]:
    """(klo, K) of FSST._truncate_frequencies (synchrosqueeze.py:91-111)."""
    klo = ctypes.c_int(0)
    K = _lib().hss_oracle_band(int(N), float(fs), float(f_lo), float(f_hi), ctypes.byref(klo))
    return klo.value, K


MODE = {"raw": 0, "abs": 1, "stack": 2}


def features(x, fs: float, window, truncate_freq: Optional[tuple] = None, mode: str = "stack",
             nthreads: int = 1, return_halfdist: bool = False):
    """Whole reference call FSST.__call__ (synchrosqueeze.py:37-65) on a batch (B, n) of float32
    windows.  Returns float32 (B, n, 2K) for "stack", (B, n, K) for "abs", complex64 (B, K, n) raw."""
    X = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    if X.ndim == 1:
        X = X[None, :]
    B, n = X.shape
    w = np.ascontiguousarray(window, dtype=np.float64).ravel()
    N = w.size
    if truncate_freq:
        klo, K = band(N, fs, truncate_freq[0], truncate_freq[1])
        has, lo, hi = 1, float(truncate_freq[0]), float(truncate_freq[1])
    else:
        K, has, lo, hi = N // 2 + 1, 0, 0.0, 0.0
    m = MODE[mode]
    if m == 1:
        out = np.empty((B, n, K), dtype=np.float32)
    elif m == 2:
        out = np.empty((B, n, 2 * K), dtype=np.float32)
    else:
        out = np.empty((B, K, n, 2), dtype=np.float32)
    hd = np.empty((B, n), dtype=np.float64) if return_halfdist else None
    if out.size:
        rc = _lib().hss_oracle_features(_fptr(X), B, n, float(fs), _dptr(w), N, has, lo, hi, m,
                                        _fptr(out), _dptr(hd) if hd is not None else None,
                                        int(nthreads))
        if rc != 0:
            raise RuntimeError(f"hss_oracle_features failed: {rc}")
    if m == 0:
        out = out.view(np.complex64)[..., 0]
    if return_halfdist:
        return out, hd
    return out


def update_mean(m: float, x: float, k: int) -> float:
    return _lib().hss_oracle_update_mean(float(m), float(x), int(k))


def update_variance(x: float, m: float, var: float, k: int) -> float:
    return _lib().hss_oracle_update_variance(float(x), float(m), float(var), int(k))


def max_threads() -> int:
    return int(_lib().hss_oracle_max_threads())
