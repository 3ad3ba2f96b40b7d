"""``ssq``-shaped shim: the native boundary the reference binds (/root/reference/hss/transforms/synchrosqueeze.py:4,48
``s, f, t = ssq.fsst(x.numpy(), self.fs, self.window)``; /root/reference/scripts/visualize_signals.py:14), served by the
HIP path.  With

    import sys, heart_sounds_segmentation_amd.ssq as ssq
    sys.modules["ssq"] = ssq

the reference's own ``hss/transforms/synchrosqueeze.py`` runs unmodified: it receives the full one-sided spectrum
(RAW mode of the C ABI) and applies its torch epilogue itself.  (The faster route is the drop-in ``FSST`` class, which
fuses that epilogue on the device; INTEGRATION.md.)

Shapes / dtypes as the reference consumes them: ``s`` complex ``(nf, nt)`` with ``nf = len(window)//2 + 1`` and
``nt = len(x)``, ``f`` float64 ``(nf,)`` = ``k fs / N`` (Nyquist row exactly ``fs/2``), ``t`` float64 ``(nt,)`` =
``arange(nt) / fs``.  ``x`` may be float32 or float64, ``(n,)`` or ``(n, 1)``; the kernels compute in float32, so a float64
input is rounded to float32 first (inside the 1e-4 tolerance of BASELINE.json; the reference casts ``s`` to complex64
right after the call, synchrosqueeze.py:51).
"""
from __future__ import annotations

import numpy as np
import torch

from .transforms.synchrosqueeze import FSST

_cache = {}


def fsst(x, fs, window):
    w = np.ascontiguousarray(window, dtype=np.float64).ravel()
    key = (float(fs), w.tobytes())
    tf = _cache.get(key)
    if tf is None:
        if len(_cache) > 16:
            _cache.clear()
        tf = _cache[key] = FSST(float(fs), w)
    xv = np.asarray(x)
    if xv.ndim == 2 and 1 in xv.shape:
        xv = xv.reshape(-1)
    if xv.ndim != 1:
        raise ValueError(f"ssq.fsst: expected a vector, got shape {xv.shape}")
    s = tf(torch.from_numpy(np.ascontiguousarray(xv))).numpy().astype(np.complex128)
    N, nt = w.size, xv.size
    f = np.arange(N // 2 + 1, dtype=np.float64) * (float(fs) / N)
    if N % 2 == 0:
        f[N // 2] = float(fs) / 2.0
    t = np.arange(nt, dtype=np.float64) / float(fs)
    return s, f, t
