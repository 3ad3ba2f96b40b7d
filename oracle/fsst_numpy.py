"""Second, independent restatement of the published fsst algorithm in numpy/scipy (fp64).

TEST INFRASTRUCTURE ONLY.  Written separately from oracle/fsst_oracle.c (library FFT instead of the
hand-rolled radix-2, scipy's not-a-knot CubicSpline instead of the dense slope solve, vectorised
``np.add.at`` scatter instead of the per-column loop) so that agreement between the two is
evidence that each follows the algorithm rather than sharing a bug.  Steps cite the same sources
as the C file: call site /root/reference/hss/transforms/synchrosqueeze.py:48; algorithm = MATLAB
fsst(x, fs, window) as published (reference README.md:5-6 names it as the origin of libssq).
"""
from __future__ import annotations

import numpy as np


def dtwin(window: np.ndarray, fs: float) -> np.ndarray:
    from scipy.interpolate import CubicSpline

    w = np.asarray(window, dtype=np.float64).ravel()
    n = w.size
    knots = np.arange(1, n + 1, dtype=np.float64)
    cs = CubicSpline(knots, w, bc_type="not-a-knot")
    return cs.derivative()(knots) * fs / (2.0 * np.pi)


def freq_vector(N: int, fs: float) -> np.ndarray:
    """psdfreqvec('npts', N, 'Fs', fs), two-sided."""
    res = fs / N
    f = res * np.arange(N, dtype=np.float64)
    if N % 2 == 0:
        f[N // 2] = fs / 2.0
    if N > 1:
        f[N - 1] = fs - res
    return f


def fsst(x, fs: float, window):
    x = np.asarray(x, dtype=np.float64).ravel()
    w = np.asarray(window, dtype=np.float64).ravel()
    nx, N = x.size, w.size
    m = N // 2
    dw = dtwin(w, fs)
    xp = np.concatenate([np.zeros(m), x, np.zeros(N - 1 - m)])
    frames = np.lib.stride_tricks.sliding_window_view(xp, N)          # (nx, N), hop 1
    V = np.fft.fft(frames * w[None, :], axis=1).T                     # (N, nx)
    Vd = np.fft.fft(frames * dw[None, :], axis=1).T
    with np.errstate(divide="ignore", invalid="ignore"):
        fcorr = -np.imag(Vd / V)
    fcorr[~np.isfinite(fcorr)] = 0.0
    fk = freq_vector(N, fs)
    finst = fk[:, None] + fcorr
    ez = np.exp(-1j * 2.0 * np.pi * m * np.arange(N) / N)
    Vm = V * ez[:, None]
    fmin, fmax = fk[0], fk[-1]
    coord = (finst - fmin) * (N - 1) / (fmax - fmin)
    r = np.sign(coord) * np.floor(np.abs(coord) + 0.5)                # MATLAB round: half away
    rows = np.mod(r, N).astype(np.int64)
    S = np.zeros((N, nx), dtype=np.complex128)
    cols = np.broadcast_to(np.arange(nx)[None, :], rows.shape)
    np.add.at(S, (rows.ravel(), cols.ravel()), Vm.ravel())
    nf = N // 2 + 1
    return S[:nf], fk[:nf].copy(), np.arange(nx, dtype=np.float64) / fs
