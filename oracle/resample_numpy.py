"""CPU restatement of the reference's Resample transform (/root/reference/hss/transforms/resample.py:13-21), i.e. of
scipy.signal.resample(x, num) for a real 1-D input, window=None (scipy >= 1.11; the reference's pixi.lock pins 1.17.0).

TEST INFRASTRUCTURE ONLY.  Written with numpy's rfft / irfft (not scipy.signal) so that it is a second implementation;
PINNED: tests/golden/resample.npz holds outputs of the reference's own Resample class run in the build container
(tests/golden/make_golden.py), and tests/test_oracle.py checks this file against them.
"""
from __future__ import annotations

import numpy as np


def resample(x, num: int) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64).ravel()
    nx = x.size
    X = np.fft.rfft(x)
    Y = np.zeros(num // 2 + 1, dtype=np.complex128)
    N = min(num, nx)
    nyq = N // 2 + 1
    Y[:nyq] = X[:nyq]
    if N % 2 == 0:
        if num < nx:          # downsampling: the copied Nyquist bin stands for both +N/2 and -N/2
            Y[N // 2] *= 2.0
        elif nx < num:        # upsampling: it is split between +N/2 and -N/2
            Y[N // 2] *= 0.5
    return np.fft.irfft(Y, num) * (float(num) / float(nx))
