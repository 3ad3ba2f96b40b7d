"""Seeded synthetic PCG inputs (the real Springer corpus needs a network download,
/root/reference/hss/datasets/heart_sounds.py:136).  Shapes follow BASELINE.md section 4.

numpy only: usable by bench.py, the tests and the golden-fixture generator alike.
"""
from __future__ import annotations

import numpy as np

SEED = 20260929


def pcg_windows(batch: int, n: int = 2000, fs: float = 1000.0, seed: int = SEED) -> np.ndarray:
    """PCG-like windows: per 0.8 s cardiac cycle (heart rate jittered +-10 %) two Gaussian-gated
    bursts, a 50 Hz "S1" and a 70 Hz "S2" (sigma ~ 20 ms), random phase per window, plus
    N(0, 0.05^2) noise; scaled so that max|x| <= 1.  Returns float32 (batch, n)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / fs
    X = np.empty((batch, n), dtype=np.float32)
    for b in range(batch):
        period = 0.8 * (1.0 + 0.1 * rng.uniform(-1.0, 1.0))
        t0 = rng.uniform(0.0, period)
        ph1, ph2 = rng.uniform(0.0, 2.0 * np.pi, size=2)
        x = np.zeros(n, dtype=np.float64)
        c = t0 - period
        while c < t[-1] + period:
            x += np.exp(-0.5 * ((t - c) / 0.020) ** 2) * np.cos(2.0 * np.pi * 50.0 * (t - c) + ph1)
            c2 = c + 0.3 * period
            x += 0.7 * np.exp(-0.5 * ((t - c2) / 0.020) ** 2) * np.cos(2.0 * np.pi * 70.0 * (t - c2) + ph2)
            c += period
        x += rng.normal(0.0, 0.05, size=n)
        x /= max(1.0, np.abs(x).max())
        X[b] = x.astype(np.float32)
    return X


def noise_windows(batch: int, n: int = 2000, seed: int = SEED + 1) -> np.ndarray:
    """N(0,1) windows: the worst case for reassignment spread."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((batch, n)).astype(np.float32)


def tone_window(n: int = 2000, fs: float = 1000.0, bin_index: int = 16, nfft: int = 128) -> np.ndarray:
    """On-bin tone cos(2*pi*(bin*fs/nfft)*t): known answer, energy lands in row `bin_index`."""
    t = np.arange(n, dtype=np.float64) / fs
    return np.cos(2.0 * np.pi * (bin_index * fs / nfft) * t).astype(np.float32)


def recording(T: int = 35500, fs: float = 1000.0, seed: int = SEED + 2) -> np.ndarray:
    """One synthetic recording (config C1): a long PCG-like signal, float32 (T,)."""
    return pcg_windows(1, n=T, fs=fs, seed=seed)[0]


def kaiser_window(N: int = 128, beta: float = 0.5) -> np.ndarray:
    """The reference's analysis window, scipy.signal.get_window(("kaiser", 0.5), 128,
    fftbins=False) (/root/reference/main.py:155) (equal to numpy.kaiser(128, 0.5) within 1 ulp; symmetric)."""
    return np.kaiser(N, beta).astype(np.float64)
