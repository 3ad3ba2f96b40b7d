"""Caller-side framing that defines the window set of the path: same semantics as the reference's
``frame_signal`` (/root/reference/hss/utils/preprocess.py:7-58) as used by the dataset loop
(/root/reference/hss/datasets/heart_sounds.py:161-168), restated for BATCHED use: one strided view
of the recording instead of a Python list of slices, so a whole recording goes to ``FSST.batch``
in one call.  Pure indexing -- no arithmetic lives here.
"""
from __future__ import annotations

from math import floor
from typing import Tuple

import numpy as np
import torch


def frame_starts(T: int, stride: int, n: int) -> Tuple[np.ndarray, np.ndarray]:
    """Start index and length of every frame ``frame_signal`` emits for a length-T signal:
    ``L = floor((T - n) / stride)`` frames ``[i*stride, i*stride + n)`` (one fewer than would fit --
    reproduced, not "fixed"); if ``L <= 0`` a single frame ``x[:n]`` (shorter when T < n)."""
    L = floor((T - n) / stride)
    if L <= 0:
        return np.zeros(1, dtype=np.int64), np.asarray([min(n, T)], dtype=np.int64)
    return np.arange(L, dtype=np.int64) * stride, np.full(L, n, dtype=np.int64)


def frame_batch(x: torch.Tensor, stride: int = 1000, n: int = 2000) -> torch.Tensor:
    """``(L, n)`` zero-copy strided view of a 1-D (or ``(T, 1)``) recording holding exactly the
    frames of ``frame_signal``; recordings shorter than ``n`` give one ``(1, T)`` frame."""
    if x.ndim == 2 and x.shape[1] == 1:
        x = x[:, 0]
    if x.ndim != 1:
        raise ValueError(f"frame_batch: expected (T,) or (T, 1), got {tuple(x.shape)}")
    T = x.shape[0]
    starts, lens = frame_starts(T, stride, n)
    if lens[0] != n:
        return x[: int(lens[0])].unsqueeze(0)
    return x.as_strided((len(starts), n), (stride * x.stride(0), x.stride(0)))
