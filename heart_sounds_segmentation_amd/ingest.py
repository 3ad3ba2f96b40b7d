"""Corpus file ingest: counterpart of ``DavidSpringerHSS._load_file``
(/root/reference/hss/datasets/heart_sounds.py:193-197), which reads ``<id>.csv`` with
``pd.read_csv(skiprows=1, names=["Signals", "Labels"])`` and returns ``(x float32 (T,), y int64 (T,))``
with labels 1..4 (1 = S1, 2 = systole, 3 = S2, 4 = diastole, README.md:15-20).  Once the transform
runs at millions of windows per second the per-recording pandas parse dominates corpus preprocessing;
this parser is a single pass in C (``hssfsst_parse_signal_csv``), no pandas needed.
"""
from __future__ import annotations

import ctypes
from typing import Tuple

import numpy as np
import torch

from . import _lib


def parse_csv_bytes(data: bytes) -> Tuple[torch.Tensor, torch.Tensor]:
    L = _lib.lib()
    n = L.hssfsst_parse_signal_csv(data, len(data), None, None, 0)
    if n < 0:
        _lib.check(int(n), "hssfsst_parse_signal_csv")
    x = np.empty(n, dtype=np.float32)
    y = np.empty(n, dtype=np.int64)
    m = L.hssfsst_parse_signal_csv(data, len(data), ctypes.c_void_p(x.ctypes.data), ctypes.c_void_p(y.ctypes.data), n)
    if m != n:
        _lib.check(int(m) if m < 0 else _lib.E_INVAL, "hssfsst_parse_signal_csv")
    return torch.from_numpy(x), torch.from_numpy(y)


def load_file(file_id: str) -> Tuple[torch.Tensor, torch.Tensor]:
    """Same contract as the reference's ``_load_file(file_id)``: reads ``file_id + ".csv"``."""
    with open(file_id + ".csv", "rb") as fh:
        return parse_csv_bytes(fh.read())
