"""Rolling (streaming) FSST for multi-channel PCG -- BASELINE.json config 5 / SURVEY section 8f row 3.

Not in the reference (which only transforms stored recordings, hss/datasets/heart_sounds.py:155-184);
built from the same path: per channel a ring of the last ``nwin - 1`` samples, every ``step`` of
``chunk`` new samples emits the ``chunk`` FSST columns whose full frames are now available (look-ahead
latency ``nwin/2 - 1`` samples) through ``hssfsst_exec_cols`` (no frame touches zero padding), and
-- because a per-signal z-score has no streaming meaning -- normalises with RUNNING mean / unbiased
std of the real and imaginary blocks, kept on the device by ``hssfsst_moments_merge`` (the chunked
form of ``hss.moments.update_mean / update_variance``, hss/moments/__init__.py:1-36).

Concatenating the un-normalised outputs of consecutive steps (zero initial ring) reproduces the
offline transform's columns ``-nwin/2 + 1, ..`` exactly: column tau of the offline, zero-padded FSST
is column ``tau + nwin/2 - 1`` of the stream.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np
import torch

from . import _lib
from .transforms.synchrosqueeze import FSST


class StreamingFSST:
    def __init__(self, channels: int, fs: float, window, truncate_freq: Optional[tuple] = None,
                 chunk: int = 128, normalize: bool = True, device: Optional[torch.device] = None):
        self.tf = FSST(fs, window, truncate_freq=truncate_freq, stack=True, device=device)
        self.nwin = int(np.asarray(window).size)
        self.channels, self.chunk, self.normalize = int(channels), int(chunk), bool(normalize)
        dev = self.tf._device_index()
        self.device = torch.device("cuda", dev)
        self.ring = torch.zeros((self.channels, self.nwin - 1), dtype=torch.float32, device=self.device)
        self.state = torch.zeros((self.channels, 6), dtype=torch.float64, device=self.device)
        self.latency_samples = self.nwin // 2 - 1

    def step(self, x_new: torch.Tensor) -> torch.Tensor:
        """``x_new``: ``(channels, chunk)`` newest samples (device tensor preferred).  Returns
        ``(channels, chunk, 2K)`` features of the columns centred ``nwin/2 - 1`` samples before the
        newest sample and earlier."""
        if tuple(x_new.shape) != (self.channels, self.chunk):
            raise ValueError(f"StreamingFSST.step: expected {(self.channels, self.chunk)}, got {tuple(x_new.shape)}")
        x_new = x_new.to(device=self.device, dtype=torch.float32)
        buf = torch.cat([self.ring, x_new], dim=1).contiguous()
        feats = self.tf.unnormalized(buf, cols=(self.nwin // 2, self.chunk))
        self.ring = buf[:, self.chunk:].contiguous()
        if self.normalize:
            L = _lib.lib()
            plan = self.tf._plan(self.device.index, _lib.MODE_STACK_UNNORM)
            stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(L.hssfsst_moments_merge(plan.handle, ctypes.c_void_p(feats.data_ptr()), self.channels,
                                               self.chunk, ctypes.c_void_p(self.state.data_ptr()), stream),
                       "hssfsst_moments_merge")
            _lib.check(L.hssfsst_normalize_running(plan.handle, ctypes.c_void_p(feats.data_ptr()), self.channels,
                                                   self.chunk, ctypes.c_void_p(self.state.data_ptr()), stream),
                       "hssfsst_normalize_running")
        return feats
