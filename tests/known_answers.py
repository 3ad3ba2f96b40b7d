"""Closed-form known answers of the synchrosqueezed transform (the published algorithm), usable
against ANY implementation through a callable ``fsst(x, fs, window) -> s`` with ``s`` complex ``(N/2+1, n)``.
They do not depend on the oracle: both the oracle (tests/test_oracle.py) and the HIP path (tests/test_gpu_parity.py)
are held against them.  They cannot replace numbers from the reference's native core (libssq is not obtainable,
DESIGN.md section 2) -- they pin the reading of the algorithm, one assumption of SURVEY appendix A each.

Each function returns ``(case name, max deviation, bound)`` tuples and asserts itself.
"""
import numpy as np


def impulse(fsst, N=128, fs=1000.0, n=600, t0=300, window=None, tol=1e-5):
    """A.3 steps 1-3, 5 (padding split, frame orientation, phase factor): a unit impulse at t0.  Frame t sees it at tap
    n' = t0 + m - t, so V[k, t] = w[n'] e^{-2 pi i k n' / N}; Vd / V = dw[n'] / w[n'] is real, the instantaneous-frequency
    correction is 0 and nothing moves: S[k, t] = w[n'] e^{-2 pi i k (n' + m) / N} for 0 <= n' < N, else 0."""
    w = np.kaiser(N, 4.0) if window is None else np.asarray(window, dtype=np.float64)
    m = N // 2
    x = np.zeros(n); x[t0] = 1.0
    s = np.asarray(fsst(x, fs, w))
    k = np.arange(N // 2 + 1)[:, None]
    t = np.arange(n)[None, :]
    tap = t0 + m - t
    inside = (tap >= 0) & (tap < N)
    want = np.where(inside, w[np.clip(tap, 0, N - 1)] * np.exp(-2j * np.pi * k * (tap + m) / N), 0.0)
    dev = np.abs(s - want).max()
    assert dev <= tol * np.abs(want).max(), ("impulse", dev)
    return ("impulse", dev)


def constant(fsst, N=128, fs=1000.0, n=500, c=2.5, tol=1e-5):
    """A.3 steps 4 and 6 at frequency zero: a constant signal is a tone at 0 Hz, and the instantaneous-frequency estimate
    of a pure tone is exact (up to the spline approximation of the derivative window), so every source bin k is sent
    to row round(k - k) = 0: in the interior S[0, t] = sum_k Vm[k, t] = N w[m] c (A.4) to within the few stray sidelobe
    cells (1e-3), the other rows hold only those strays, and the reconstruction identity holds exactly."""
    w = np.kaiser(N, 3.0)
    x = np.full(n, c)
    s = np.asarray(fsst(x, fs, w))
    inner = slice(N, n - N)
    full = N * w[N // 2] * c
    dev = np.abs(s[0, inner] - full).max()
    assert dev <= 1e-3 * abs(full), ("constant row 0", dev)
    assert np.abs(s[1:, inner]).max() <= 1e-3 * abs(full), ("constant other rows", np.abs(s[1:, inner]).max())
    rec = (s[0] + s[N // 2] + 2 * s[1:N // 2].sum(0)).real / (N * w[N // 2])
    dev2 = np.abs(rec[inner] - c).max()
    assert dev2 <= 10 * tol * abs(c), ("constant reconstruction", dev2)
    return ("constant", max(dev / abs(full), dev2))


def off_bin_tone(fsst, N=128, fs=1000.0, n=1200, bin_pos=16.3, tol=0.01):
    """A.3 steps 4 and 6 (sign of the correction, bins units, rounding): a pure tone between bins.  The instantaneous-
    frequency estimate of a pure tone is exact for any window, so EVERY source bin of the main lobe is sent to
    round(bin_pos) = 16: that row carries (almost) the whole column, its neighbours nothing."""
    from scipy.signal import get_window
    w = get_window("hann", N, fftbins=False)
    t = np.arange(n) / fs
    x = np.cos(2 * np.pi * bin_pos * fs / N * t)
    s = np.abs(np.asarray(fsst(x, fs, w)))
    inner = slice(N, n - N)
    frac = s[int(round(bin_pos)), inner] / s[:, inner].sum(0)
    assert frac.min() >= 1.0 - tol, ("off-bin tone", frac.min())
    # and the row's magnitude is that of the tone seen through the window at its true frequency: |sum w e^{i phase}| / 2
    return ("off_bin_tone", 1.0 - frac.min())


def linear_chirp(fsst, N=128, fs=1000.0, n=2000, f0=60.0, f1=260.0):
    """A.3 step 4 on a non-stationary signal: a linear chirp under a Gaussian window (the textbook case in which the
    STFT ridge sits on the instantaneous frequency): in the interior the strongest row of column t is the bin of
    f(t) = f0 + (f1 - f0) t / T, within one bin (the first-order estimate is exact on the ridge only)."""
    from scipy.signal import get_window
    w = get_window(("gaussian", N / 8.0), N, fftbins=False)
    T = n / fs
    t = np.arange(n) / fs
    x = np.cos(2 * np.pi * (f0 * t + 0.5 * (f1 - f0) / T * t * t))
    s = np.abs(np.asarray(fsst(x, fs, w)))
    inner = np.arange(N, n - N)
    ridge = s[:, inner].argmax(0)
    want = (f0 + (f1 - f0) * t[inner] / T) * N / fs
    dev = np.abs(ridge - want).max()
    assert dev <= 1.0, ("chirp ridge", dev)
    return ("linear_chirp", dev)


def shift_covariance(fsst, N=128, fs=1000.0, n=900, d=37, exact=True):
    """A.3 steps 1-2 (hop 1, centred frames): dropping the first d samples shifts the interior columns by d and changes
    nothing else -- the frames are the same vectors.  Bit-exact for an implementation whose per-frame arithmetic does
    not depend on where the frame sits (the oracle; the HIP kernels, whose frames are lanes of identical code)."""
    rng = np.random.default_rng(5)
    w = np.kaiser(N, 0.5)
    x = rng.standard_normal(n)
    a = np.asarray(fsst(x, fs, w))
    b = np.asarray(fsst(x[d:], fs, w))
    lo, hi = N, n - d - N
    dev = np.abs(a[:, lo + d:hi + d] - b[:, lo:hi]).max()
    assert (dev == 0.0) if exact else (dev <= 1e-5 * np.abs(a).max()), ("shift covariance", dev)
    return ("shift_covariance", dev)


def homogeneity(fsst, N=128, fs=1000.0, n=700):
    """The reassignment depends on Vd / V only: S(2^p x) = 2^p S(x) exactly (power-of-two scaling is exact in binary
    floating point), and S(-x) = -S(x)."""
    rng = np.random.default_rng(9)
    w = np.hanning(N)
    x = rng.standard_normal(n)
    a = np.asarray(fsst(x, fs, w))
    dev = max(np.abs(np.asarray(fsst(8.0 * x, fs, w)) - 8.0 * a).max(), np.abs(np.asarray(fsst(-x, fs, w)) + a).max())
    assert dev == 0.0, ("homogeneity", dev)
    return ("homogeneity", dev)


def odd_window(fsst, N=33, fs=1000.0, n=300, t0=150, tol=1e-5):
    """A.3 steps 1 and 5 for odd N (padding floor(N/2) in front, N-1-floor(N/2) behind; phase factor
    exp(-2 pi i floor(N/2) k / N), no Nyquist row): the impulse answer again."""
    return impulse(fsst, N=N, fs=fs, n=n, t0=t0, window=np.hamming(N), tol=tol)


ALL = (impulse, constant, off_bin_tone, linear_chirp, shift_covariance, homogeneity, odd_window)
