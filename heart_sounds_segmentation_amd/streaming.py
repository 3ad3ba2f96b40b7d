"""Rolling (streaming) FSST for multi-channel PCG -- BASELINE.json config 5 / SURVEY section 8f row 3.

Not in the reference (which only transforms stored recordings, hss/datasets/heart_sounds.py:155-184);
built from the same path: per channel the last ``nwin - 1`` samples are kept, every ``step`` of
``chunk`` new samples emits the ``chunk`` FSST columns whose full frames are now available (look-ahead
latency ``nwin/2 - 1`` samples) through ``hssfsst_exec_frames`` (no frame touches zero padding), and
-- because a per-signal z-score has no streaming meaning -- normalises with RUNNING mean / unbiased
std of the real and imaginary blocks, kept on the device by ``hssfsst_moments_merge`` (the chunked
form of ``hss.moments.update_mean / update_variance``, hss/moments/__init__.py:1-36).

Concatenating the un-normalised outputs of consecutive steps (zero initial history; ``step`` returns a fresh tensor
unless called with ``copy=False``, which hands out the object's single output buffer) reproduces the
offline transform's columns ``-nwin/2 + 1, ..`` exactly: column tau of the offline, zero-padded FSST
is column ``tau + nwin/2 - 1`` of the stream.

No allocation per step: the history lives in ONE preallocated device buffer per channel of
``nwin - 1 + slots * chunk`` samples that is written like a tape -- each step appends its chunk and
transforms the last ``nwin - 1 + chunk`` samples in place (``x_stride`` = tape length); only when the
tape is full (every ``slots`` steps) the last ``nwin - 1`` samples are copied back to its start.  The
feature output and the pinned host staging buffers of ``step_host`` are preallocated too.

A step is ONE native call, ``hssfsst_stream_step`` (include/hssfsst.h), and for window lengths 256 / 512 with an even band of
at most 24 rows (BASELINE config 5) ONE kernel launch: the kernel appends the chunk to the tape (reading it where it lies,
pinned host memory included), transforms it, forms every 16-frame group's float64 moment sums from the registers that hold its
features, and the last block of a channel merges the running moments and normalises the chunk -- into the device buffer and,
``step_host``, straight into the pinned host buffer; what remains of ``step_host`` is the one synchronisation.  Other shapes
take copy + transform + one merge-and-normalise launch; both routes give the same bits.
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
                 chunk: int = 128, normalize: bool = True, device: Optional[torch.device] = None,
                 slots: int = 64):
        self.tf = FSST(fs, window, truncate_freq=truncate_freq, stack=True, device=device)
        self.nwin = int(np.asarray(window).size)
        self.channels, self.chunk, self.normalize = int(channels), int(chunk), bool(normalize)
        dev = self.tf._device_index()
        self.device = torch.device("cuda", dev)
        self.hist = self.nwin - 1
        self.slots = max(1, int(slots))
        self.tape_len = self.hist + self.slots * self.chunk
        self.tape = torch.zeros((self.channels, self.tape_len), dtype=torch.float32, device=self.device)
        self._wrap = torch.empty((self.channels, self.hist), dtype=torch.float32, device=self.device)
        self.pos = self.hist                               # where the next chunk goes (history = zeros before it)
        self.state = torch.zeros((self.channels, 6), dtype=torch.float64, device=self.device)
        self.latency_samples = self.nwin // 2 - 1
        self._plan = self.tf._plan(dev, _lib.MODE_STACK_UNNORM)
        self.K = self._plan.K
        self.out = torch.empty((self.channels, self.chunk, 2 * self.K), dtype=torch.float32, device=self.device)
        self._pin_in = None
        self._pin_out = None
        self._keep = None
        self._pin_ring = None

    def last_kernel(self) -> str:
        """The transform kernel the last step ran (``hssfsst_plan_last_kernel`` of this stream's plan): one launch per step
        reads ``fsst_core128_kernel<..., stream, pairs> [...]``."""
        buf = ctypes.create_string_buffer(160)
        _lib.check(_lib.lib().hssfsst_plan_last_kernel(self._plan.handle, buf, len(buf)), "hssfsst_plan_last_kernel")
        return buf.value.decode()

    # kept for callers / tests that looked at the history of the first implementation
    @property
    def ring(self) -> torch.Tensor:
        return self.tape[:, self.pos - self.hist:self.pos]

    def _make_room(self) -> None:
        if self.pos + self.chunk > self.tape_len:          # tape full: history back to the start (every `slots` steps)
            self._wrap.copy_(self.tape[:, self.pos - self.hist:self.pos])     # (two hops: the ranges may overlap)
            self.tape[:, :self.hist].copy_(self._wrap)
            self.pos = self.hist

    def _native_step(self, x_ptr: int, x_stride: int, x_on_device: bool, host_out_ptr: Optional[int]) -> None:
        """One ``hssfsst_stream_step``: copy the chunk into the tape, transform, merge + normalise, and (host_out_ptr)
        copy the features to pinned host memory and wait -- 3 launches (+ the copies), one C call."""
        self._make_room()
        L = _lib.lib()
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(L.hssfsst_stream_step(self._plan.handle, ctypes.c_void_p(self.tape.data_ptr()), self.tape_len, self.pos,
                                         ctypes.c_void_p(x_ptr), x_stride, 1 if x_on_device else 0, self.channels, self.chunk,
                                         ctypes.c_void_p(self.out.data_ptr()),
                                         ctypes.c_void_p(self.state.data_ptr()) if self.normalize else None,
                                         ctypes.c_void_p(host_out_ptr) if host_out_ptr else None, stream),
                   "hssfsst_stream_step")
        self.pos += self.chunk

    def step(self, x_new: torch.Tensor, copy: bool = True) -> torch.Tensor:
        """``x_new``: ``(channels, chunk)`` newest samples (device tensor preferred).  Returns
        ``(channels, chunk, 2K)`` features of the columns centred ``nwin/2 - 1`` samples before the
        newest sample and earlier.

        ``copy=True`` (default) returns a fresh tensor, so ``[st.step(x) for x in chunks]`` or ``torch.cat`` over collected
        outputs behave as expected.  ``copy=False`` returns THIS OBJECT'S OUTPUT BUFFER, which the next step overwrites
        (no allocation per step: the latency-critical loop of BASELINE config 5).
        A host ``x_new`` is first copied into one of two pinned staging buffers of this object, so the caller may refill
        its own buffer as soon as ``step`` returns (the upload itself is asynchronous)."""
        if tuple(x_new.shape) != (self.channels, self.chunk):
            raise ValueError(f"StreamingFSST.step: expected {(self.channels, self.chunk)}, got {tuple(x_new.shape)}")
        if not x_new.is_cuda:
            if self._pin_ring is None:
                self._pin_ring = [torch.empty((self.channels, self.chunk), dtype=torch.float32).pin_memory() for _ in range(2)]
                self._pin_ev = [torch.cuda.Event() for _ in range(2)]
                self._pin_k = 0
            k = self._pin_k
            self._pin_ev[k].synchronize()                  # the upload that last read this staging buffer has finished
            self._pin_ring[k].copy_(x_new)                 # (converts dtype / layout as needed)
            self._native_step(self._pin_ring[k].data_ptr(), self.chunk, False, None)
            self._pin_ev[k].record(torch.cuda.current_stream(self.device))
            self._pin_k = k ^ 1
            return self.out.clone() if copy else self.out
        if x_new.dtype != torch.float32 or x_new.stride(1) != 1 or x_new.stride(0) < self.chunk:
            x_new = x_new.to(torch.float32).contiguous()   # (a column slice of a wider row-major tensor is taken as it is)
        if x_new.device != self.device:
            x_new = x_new.to(self.device)
        self._native_step(x_new.data_ptr(), x_new.stride(0), True, None)
        self._keep = x_new                                 # the copy is asynchronous: keep the source alive until the next step
        return self.out.clone() if copy else self.out

    def step_unfused(self, x_new: torch.Tensor) -> torch.Tensor:
        """The same step as separate calls (copy, ``hssfsst_exec_frames``, ``hssfsst_moments_merge``,
        ``hssfsst_normalize_running``): what ``step`` did before ``hssfsst_stream_step`` existed; kept for the
        bit-equality test of the two routes."""
        if tuple(x_new.shape) != (self.channels, self.chunk):
            raise ValueError(f"StreamingFSST.step: expected {(self.channels, self.chunk)}, got {tuple(x_new.shape)}")
        self._make_room()
        self.tape[:, self.pos:self.pos + self.chunk].copy_(x_new, non_blocking=True)
        L = _lib.lib()
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        n_in = self.hist + self.chunk
        x0 = self.tape.data_ptr() + 4 * (self.pos - self.hist)
        _lib.check(L.hssfsst_exec_frames(self._plan.handle, ctypes.c_void_p(x0), self.channels, n_in, self.tape_len,
                                         self.nwin // 2, self.chunk, 1, ctypes.c_void_p(self.out.data_ptr()), 1, stream),
                   "hssfsst_exec_frames")
        self.pos += self.chunk
        if self.normalize:
            _lib.check(L.hssfsst_moments_merge(self._plan.handle, ctypes.c_void_p(self.out.data_ptr()), self.channels,
                                               self.chunk, ctypes.c_void_p(self.state.data_ptr()), stream),
                       "hssfsst_moments_merge")
            _lib.check(L.hssfsst_normalize_running(self._plan.handle, ctypes.c_void_p(self.out.data_ptr()), self.channels,
                                                   self.chunk, ctypes.c_void_p(self.state.data_ptr()), stream),
                       "hssfsst_normalize_running")
        return self.out

    def step_host(self, x_new: np.ndarray) -> np.ndarray:
        """Host in, host out (the latency BASELINE config 5 asks for: last sample of a chunk on the host ->
        its features on the host): pinned staging buffers and ONE native call -- H2D copy straight into the tape, the
        kernels, D2H copy, one synchronisation.  Returns a view of the pinned output buffer (overwritten by the next
        call)."""
        if self._pin_in is None:
            self._pin_in = torch.empty((self.channels, self.chunk), dtype=torch.float32).pin_memory()
            self._pin_out = torch.empty((self.channels, self.chunk, 2 * self.K), dtype=torch.float32).pin_memory()
            self._pin_out_np = self._pin_out.numpy()
            self._pin_in_np = self._pin_in.numpy()
        self._pin_in_np[...] = x_new
        self._native_step(self._pin_in.data_ptr(), self.chunk, False, self._pin_out.data_ptr())
        return self._pin_out_np
