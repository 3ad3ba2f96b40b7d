"""Mirror of ``hss.transforms.Resample`` (/root/reference/hss/transforms/resample.py:5-21).

The reference's transform is ``torch.tensor(scipy.signal.resample(x.cpu(), self.num), dtype=dtype)``: Fourier-method
resampling to a fixed number of samples.  The dataset applies it to the label vector of a recording when a
``Resample`` is found in the transform chain (hss/datasets/heart_sounds.py:202-207: ``round(t(y)) - 1``).  Here the
arithmetic is the library's host helper ``hssfsst_resample`` (csrc/fourier_resample.hpp, fp64, any length); scipy is
not imported.  Same constructor, same call signature, same output type (a fresh CPU tensor of ``dtype``).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from .. import _lib


class Resample:
    def __init__(self, num: int) -> None:
        """
        Args:
            num (int): number of output samples
        """
        self.num = num

    def __call__(self, x: torch.Tensor, dtype: torch.dtype = torch.float32) -> torch.Tensor:
        xs = x.detach().cpu() if isinstance(x, torch.Tensor) else torch.as_tensor(x)
        if xs.dim() != 1:
            # scipy resamples along axis 0; the reference only ever passes 1-D signals / label vectors
            if xs.dim() == 2 and xs.shape[1] == 1:
                return self(xs[:, 0], dtype).unsqueeze(1)
            raise ValueError(f"Resample expects a 1-D tensor, got shape {tuple(xs.shape)}")
        if xs.is_complex():
            raise ValueError("Resample: complex input is not supported")
        xin = np.ascontiguousarray(xs.to(torch.float64).numpy())
        num = int(self.num)
        if num < 1 or xin.size < 1:
            raise ValueError(f"Resample: need at least one input and one output sample (n={xin.size}, num={num})")
        y = np.empty(num, dtype=np.float64)
        dp = ctypes.POINTER(ctypes.c_double)
        rc = _lib.lib().hssfsst_resample(xin.ctypes.data_as(dp), xin.size, num, y.ctypes.data_as(dp))
        _lib.check(rc, "hssfsst_resample")
        return torch.from_numpy(y).to(dtype)


def resample_labels(y: torch.Tensor, t: Resample) -> torch.Tensor:
    """The label rule of DavidSpringerHSS._apply_transform (hss/datasets/heart_sounds.py:205-206)."""
    return torch.round(t(y)).type(torch.int64) - 1
