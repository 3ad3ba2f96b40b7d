"""``FSST`` -- drop-in for the reference's ``hss.transforms.FSST``
(/root/reference/hss/transforms/synchrosqueeze.py:8-111) backed by the gfx950 HIP kernels behind
the C ABI of ``include/hssfsst.h``.

Same constructor, attribute names, call signature, output shapes / dtypes / orientation and error
behaviour as the reference class; the body (native ``ssq.fsst`` + torch epilogue) is replaced by one
``hssfsst_exec`` call.  Extensions that the reference does not have: ``device=`` and ``batch()``.
There is no CPU implementation here: without the HIP library or a GPU every call raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import numpy as np
import numpy.typing as npt
import torch

from .. import _lib


def _release_pinned(plan, addr: int) -> None:
    """A pinned pool buffer goes back to its plan (the borrower holds the plan alive until then)."""
    try:
        if plan.handle and plan.handle.value and plan.pid == os.getpid():
            plan._L.hssfsst_pinned_release(plan.handle, ctypes.c_void_p(addr))
    except Exception:
        pass


_LENT_TYPES = {}


def _lent_type(nfl: int):
    """ctypes float array type over a LENT pinned pool buffer: the numpy array / tensor made from an instance keeps it alive, and when the last
    view of the result is gone the instance's ``__del__`` hands the buffer back (``hssfsst_pinned_release``).  (A subclass with ``__del__``
    + ``np.frombuffer`` + ``torch.from_numpy`` costs 1.4 us per call; ``weakref.finalize`` + ``torch.frombuffer`` + ``view`` cost 3.3.)"""
    T = _LENT_TYPES.get(nfl)
    if T is None:
        class _Lent(ctypes.c_float * nfl):
            _plan = None

            def __del__(self):
                _release_pinned(self._plan, ctypes.addressof(self))
        T = _LENT_TYPES[nfl] = _Lent
    return T


class _Plan:
    """Owner of one ``hssfsst_plan*`` (created lazily in the calling process: fork-safe)."""

    def __init__(self, device_index: int, window: np.ndarray, fs: float, band, mode: int):
        _lib.guard_fork()
        L = _lib.lib()
        self._L = L
        self.handle = ctypes.c_void_p()
        w = np.ascontiguousarray(window, dtype=np.float64).ravel()
        has_band = 1 if band else 0
        lo, hi = (float(band[0]), float(band[1])) if band else (0.0, 0.0)
        rc = L.hssfsst_plan_create(ctypes.byref(self.handle), int(device_index), int(w.size),
                                   w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), float(fs),
                                   has_band, lo, hi, int(mode))
        _lib.check(rc, "hssfsst_plan_create")
        vals = [ctypes.c_int() for _ in range(7)]
        _lib.check(L.hssfsst_plan_info(self.handle, *[ctypes.byref(v) for v in vals]), "hssfsst_plan_info")
        self.nwin, self.nf, self.klo, self.K, self.ofps, self.mode, self.device = [v.value for v in vals]
        self.pid = os.getpid()

    def __del__(self):
        try:
            if getattr(self, "handle", None) and self.handle.value and self.pid == os.getpid():
                self._L.hssfsst_plan_destroy(self.handle)
        except Exception:
            pass


class FSST:
    """
    Fourier Synchrosqueezed Transform (MI355X / HIP implementation).

    Args mirror the reference constructor (synchrosqueeze.py:13-35):
        fs: sample frequency.
        window: analysis window (numpy array); its length is the FFT length.
        abs: return ``abs(s).t()`` -> float32 ``(n, K)``.
        stack: return the z-scored real and imaginary parts stacked -> float32 ``(n, 2K)``.
        truncate_freq: ``(lo, hi)`` in Hz, inclusive; keeps K rows.  None/empty keeps all.
        dtype: kept for signature parity (the reference only uses it for ``f``/``t``).
        device (extension): torch device for the computation; default ``cuda`` (current device).
    Precedence as in ``__call__`` (synchrosqueeze.py:56-65): truncate, then ``abs`` wins over
    ``stack``, else the raw complex64 ``(K, n)`` spectrum.
    """

    def __init__(
        self,
        fs: float,
        window: npt.NDArray,
        abs: bool = False,
        stack: bool = False,
        truncate_freq: Optional[tuple] = None,
        dtype: torch.dtype = torch.float32,
        device: Optional[torch.device] = None,
    ):
        self.fs: float = fs
        self.window: npt.NDArray = window
        self.abs = abs
        self.stack = stack
        self.truncate_freq = truncate_freq
        self.dtype = dtype
        self.device = device
        self._plans = {}
        self._cuda_seen = False

    # ------------------------------------------------------------------ plan / geometry
    def _mode(self) -> int:
        if self.abs:
            return _lib.MODE_ABS
        if self.stack:
            return _lib.MODE_STACK
        return _lib.MODE_RAW

    def _device_index(self, like: Optional[torch.Tensor] = None) -> int:
        _lib.guard_fork()
        if like is not None and like.is_cuda:
            return like.device.index if like.device.index is not None else torch.cuda.current_device()
        if not torch.cuda.is_available():
            raise RuntimeError("FSST: no HIP device visible (torch.cuda.is_available() is False); "
                               "this implementation has no CPU fallback")
        if self.device is not None:
            d = torch.device(self.device)
            if d.type != "cuda":
                raise RuntimeError(f"FSST: device {d} is not a HIP device; there is no CPU path")
            return d.index if d.index is not None else torch.cuda.current_device()
        return torch.cuda.current_device()

    def _plan(self, device_index: int, mode: Optional[int] = None) -> _Plan:
        mode = self._mode() if mode is None else mode
        key = (os.getpid(), device_index, mode)
        plan = self._plans.get(key)
        if plan is None:
            band = tuple(self.truncate_freq) if self.truncate_freq else None
            plan = _Plan(device_index, np.asarray(self.window), float(self.fs), band, mode)
            self._plans[key] = plan
        return plan

    def __getstate__(self):          # plans hold device handles: never pickle them into workers
        st = self.__dict__.copy()
        st["_plans"] = {}
        st["_cuda_seen"] = False
        return st

    def band(self):
        """(klo, K): first kept row and number of kept rows (host-side; no GPU needed)."""
        klo, K = ctypes.c_int(), ctypes.c_int()
        w = np.asarray(self.window)
        if self.truncate_freq:
            _lib.check(_lib.lib().hssfsst_band(int(w.size), float(self.fs), float(self.truncate_freq[0]),
                                               float(self.truncate_freq[1]), ctypes.byref(klo),
                                               ctypes.byref(K)), "hssfsst_band")
            return klo.value, K.value
        return 0, int(w.size) // 2 + 1

    # ------------------------------------------------------------------ execution
    def _run(self, X: torch.Tensor, mode: Optional[int] = None,
             out: Optional[torch.Tensor] = None, cols: Optional[tuple] = None) -> torch.Tensor:
        """X: (B, n) float32, CPU or cuda, unit stride along n and a positive stride between signals (a dense
        batch, or overlapping frames of one recording as made by framing.frame_batch -- read in place).
        Returns per-mode tensor on X's device.  cols = (col0, ncols) restricts the output to those frame centres."""
        B, n_in = X.shape
        xstride = int(X.stride(0)) if B > 1 else int(n_in)
        if n_in > 1 and X.stride(1) != 1:
            raise ValueError("FSST: samples of a signal must be contiguous")
        col0, n = cols if cols is not None else (0, n_in)
        if col0 < 0 or n < 1 or col0 + n > n_in:
            raise ValueError(f"FSST: column range ({col0}, {n}) outside a signal of {n_in} samples")
        if n_in < 1:
            raise ValueError("FSST: empty signal")
        dev = self._device_index(X)
        plan = self._plan(dev, mode)
        K = plan.K
        m = plan.mode
        on_dev = X.is_cuda
        odev = X.device
        if m == _lib.MODE_RAW:
            shape, dt = (B, K, n), torch.complex64
        elif m == _lib.MODE_ABS:
            shape, dt = (B, n, K), torch.float32
        else:
            shape, dt = (B, n, 2 * K), torch.float32
        if out is None:
            out = torch.empty(shape, dtype=dt, device=odev)
        elif (tuple(out.shape) != shape or out.dtype != dt or out.device != odev
              or not out.is_contiguous()):
            raise ValueError(f"FSST: out must be a contiguous {dt} tensor of shape {shape} on {odev}")
        if B == 0 or K == 0:
            return out
        stream = torch.cuda.current_stream(dev).cuda_stream if on_dev else None
        rc = _lib.lib().hssfsst_exec_frames(plan.handle, ctypes.c_void_p(X.data_ptr()), int(B), int(n_in), xstride,
                                            int(col0), int(n), 1 if on_dev else 0,
                                            ctypes.c_void_p(out.data_ptr()), 1 if on_dev else 0,
                                            ctypes.c_void_p(stream) if stream else None)
        _lib.check(rc, "hssfsst_exec_frames")
        return out

    @staticmethod
    def _as_f32(x: torch.Tensor, frames: bool = False) -> torch.Tensor:
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.asarray(x))
        if x.is_complex():
            raise ValueError("FSST: real input expected")
        x = x.detach().to(torch.float32)
        # frames=True: a (B, n) view with unit stride along n and a positive signal stride is read in place
        # (overlapping frames of one recording: no duplicated copy); anything else is made dense
        if frames and x.ndim == 2 and x.shape[1] > 1 and x.stride(1) == 1 and (x.shape[0] <= 1 or x.stride(0) >= 1):
            return x
        return x.contiguous()

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        """
        Computes the transform of ONE signal, like the reference ``__call__``
        (synchrosqueeze.py:37-65).  Accepts what the reference's callers pass: ``(n,)`` or ``(n, 1)``
        float32 (dataset frames, heart_sounds.py:167,181) or float64 (visualisation script).
        A CPU tensor returns a CPU tensor, a cuda tensor (extension) stays on the device.

        Returns: ``stack`` -> float32 ``(n, 2K)``; ``abs`` -> float32 ``(n, K)``; otherwise
        complex64 ``(K, n)``.
        """
        # the dataset loop's call -- a CPU float32 (n,) / (n, 1) frame, once per 2000 samples (heart_sounds.py:166-168,199-201) --
        # goes straight to the C ABI: of a 0.057 ms call the generic path below spent 0.012 ms on conversions and checks
        if (type(x) is torch.Tensor and x.dtype is torch.float32 and x.is_cpu and x.is_contiguous()
                and not x.requires_grad and (x.dim() == 1 or (x.dim() == 2 and 1 in x.shape)) and x.numel() > 0):
            if self.device is None and self._cuda_seen:  # (the checks of _device_index were made by an earlier call of this process)
                _lib.guard_fork()
                plan = self._plan(torch.cuda.current_device())
            else:
                plan = self._plan(self._device_index(x))
                self._cuda_seen = True
            n, K, m = x.numel(), plan.K, plan.mode
            if K > 0:
                L = _lib.lib()
                # the kernels store the features into a pinned buffer of the plan's pool and the returned tensor IS that buffer (no 352 kB
                # copy): it goes back to the pool when the last view of the result dies; a caller that keeps every result finds the pool lent
                # out after 64 frames and gets freshly allocated tensors filled by a copy, as before (hssfsst.h: hssfsst_exec_pinned)
                ptr = ctypes.c_void_p()
                rc = L.hssfsst_exec_pinned(plan.handle, x.data_ptr(), n, ctypes.byref(ptr))
                if rc == 0:
                    buf = _lent_type(n * plan.ofps).from_address(ptr.value)
                    buf._plan = plan
                    if m == _lib.MODE_RAW:
                        return torch.from_numpy(np.frombuffer(buf, dtype=np.complex64).reshape(K, n))
                    return torch.from_numpy(np.frombuffer(buf, dtype=np.float32).reshape(n, K if m == _lib.MODE_ABS else 2 * K))
                if rc < 0:
                    _lib.check(rc, "hssfsst_exec_pinned")
                out = (torch.empty((K, n), dtype=torch.complex64) if m == _lib.MODE_RAW
                       else torch.empty((n, K if m == _lib.MODE_ABS else 2 * K), dtype=torch.float32))
                _lib.check(L.hssfsst_exec_frames(plan.handle, x.data_ptr(), 1, n, n, 0, n, 0, out.data_ptr(), 0, None),
                           "hssfsst_exec_frames")
                return out
        x = self._as_f32(x)
        if x.ndim == 2 and 1 in x.shape:
            x = x.reshape(-1)
        if x.ndim != 1:
            raise ValueError(f"FSST: expected a single signal of shape (n,) or (n, 1), got "
                             f"{tuple(x.shape)}; use FSST.batch for (B, n)")
        return self._run(x.unsqueeze(0))[0]

    def batch(self, X: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Extension: transform ``B`` independent signals at once.  ``X``: ``(B, n)`` (CPU or cuda).
        Returns ``(B, n, 2K)`` / ``(B, n, K)`` / complex64 ``(B, K, n)`` on X's device; a CPU input
        is staged through the device by the library.  ``out`` optionally receives the result."""
        X = self._as_f32(X, frames=True)
        if X.ndim == 3 and X.shape[-1] == 1:
            X = X[..., 0]
        if X.ndim != 2:
            raise ValueError(f"FSST.batch: expected (B, n), got {tuple(X.shape)}")
        return self._run(X, out=out)

    def frames(self, x: torch.Tensor, starts, n: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Extension (batched dataset builder): transform the frames
        ``x[starts[b] : starts[b] + n]`` of ONE 1-D buffer -- e.g. many recordings laid back to back, ``starts`` =
        every frame of every recording (``framing.frame_starts``) -- in one call (``hssfsst_exec_list``).
        ``x``: ``(T,)`` float32, CPU or cuda; ``starts``: int64 sequence / tensor (CPU, or on x's device).
        Returns ``(len(starts), n, 2K)`` / ``(.., n, K)`` / complex64 ``(.., K, n)`` on x's device."""
        x = self._as_f32(x)
        if x.ndim != 1:
            raise ValueError(f"FSST.frames: expected one 1-D buffer, got {tuple(x.shape)}")
        if not isinstance(starts, torch.Tensor):
            starts = torch.as_tensor(np.asarray(starts, dtype=np.int64))
        starts = starts.to(torch.int64).contiguous()
        if starts.ndim != 1:
            raise ValueError("FSST.frames: starts must be 1-D")
        B, T, n = int(starts.shape[0]), int(x.shape[0]), int(n)
        if n < 1 or T < n:
            raise ValueError(f"FSST.frames: frame length {n} does not fit a buffer of {T} samples")
        if starts.is_cuda and (not x.is_cuda or starts.device != x.device):
            starts = starts.cpu()
        if B and not starts.is_cuda and (int(starts.min()) < 0 or int(starts.max()) > T - n):
            raise ValueError(f"FSST.frames: a frame start lies outside [0, {T - n}]")
        dev = self._device_index(x)
        plan = self._plan(dev)
        K, m = plan.K, plan.mode
        if m == _lib.MODE_RAW:
            shape, dt = (B, K, n), torch.complex64
        elif m == _lib.MODE_ABS:
            shape, dt = (B, n, K), torch.float32
        else:
            shape, dt = (B, n, 2 * K), torch.float32
        if out is None:
            out = torch.empty(shape, dtype=dt, device=x.device)
        elif tuple(out.shape) != shape or out.dtype != dt or out.device != x.device or not out.is_contiguous():
            raise ValueError(f"FSST.frames: out must be a contiguous {dt} tensor of shape {shape} on {x.device}")
        if B == 0 or K == 0:
            return out
        on_dev = x.is_cuda
        stream = torch.cuda.current_stream(dev).cuda_stream if on_dev else None
        rc = _lib.lib().hssfsst_exec_list(plan.handle, ctypes.c_void_p(x.data_ptr()), T, ctypes.c_void_p(starts.data_ptr()),
                                          1 if starts.is_cuda else 0, B, n, 1 if on_dev else 0,
                                          ctypes.c_void_p(out.data_ptr()), 1 if on_dev else 0,
                                          ctypes.c_void_p(stream) if stream else None)
        _lib.check(rc, "hssfsst_exec_list")
        return out

    def check(self, device_index: Optional[int] = None) -> int:
        """Extension: waits for the device and raises ``RuntimeError`` if a kernel reported a failed internal wait
        (the library also reports it at the start of the plan's next call, without being asked).  Returns the path of
        the plan's last ``stack`` call: 0 = two launches, 1 = the one-CU-per-signal z-score kernel (when preferred),
        2 = the team kernel (features normalised in registers, written once: the default for the reference's band and
        signals up to 2048 samples); truthy = a single launch."""
        dev = self._device_index() if device_index is None else device_index
        plan = self._plan(dev)
        _lib.check(_lib.lib().hssfsst_plan_check(plan.handle), "hssfsst_plan_check")
        return int(_lib.lib().hssfsst_plan_last_exec_fused(plan.handle))

    def last_kernel(self, device_index: Optional[int] = None) -> str:
        """Extension: the transform kernel that device's plan ran last -- ``"<instantiation> [<waves> waves/block, grid <n>]"``
        (which of the library's kernels a window length / band / mode / shape takes is otherwise only visible in a trace)."""
        dev = self._device_index() if device_index is None else device_index
        buf = ctypes.create_string_buffer(160)
        _lib.check(_lib.lib().hssfsst_plan_last_kernel(self._plan(dev).handle, buf, len(buf)), "hssfsst_plan_last_kernel")
        return buf.value.decode()

    def set_zpath(self, zpath: str = "auto", device_index: Optional[int] = None) -> None:
        """Extension: preference among the z-score paths of the following ``stack`` calls on that device's plan --
        "auto" (the fastest that applies), "two_launch", "one_cu", "team".  Results do not depend on it (bit-identical)."""
        dev = self._device_index() if device_index is None else device_index
        code = {"auto": 0, "two_launch": 1, "one_cu": 2, "team": 3}[zpath]
        _lib.check(_lib.lib().hssfsst_plan_set_zpath(self._plan(dev).handle, code), "hssfsst_plan_set_zpath")

    def fallbacks(self, device_index: Optional[int] = None) -> int:
        """Extension: how many team-kernel launches of that device's plan gave themselves up and were computed by the
        two-launch kernels queued behind them (other processes kept the team's blocks apart): a performance event, not
        an error.  Call after a synchronisation."""
        dev = self._device_index() if device_index is None else device_index
        return int(_lib.lib().hssfsst_plan_fallbacks(self._plan(dev).handle))

    def set_timing(self, enable, device_index: Optional[int] = None) -> None:
        """Extension (bench): record HIP events around the kernels of every following call (``True`` / 1) or of every n-th one
        (an int n > 1: the events themselves cost stream time); ``False`` / 0: off."""
        dev = self._device_index() if device_index is None else device_index
        _lib.check(_lib.lib().hssfsst_plan_set_timing(self._plan(dev).handle, int(enable)),
                   "hssfsst_plan_set_timing")

    def timing(self, device_index: Optional[int] = None):
        """(core_ms_sum, normalize_ms_sum, n_calls) since ``set_timing(True)``; synchronises."""
        dev = self._device_index() if device_index is None else device_index
        ms = (ctypes.c_float * 2)()
        cnt = ctypes.c_int()
        _lib.check(_lib.lib().hssfsst_plan_timing(self._plan(dev).handle, ms, ctypes.byref(cnt)),
                   "hssfsst_plan_timing")
        return float(ms[0]), float(ms[1]), cnt.value

    def unnormalized(self, X: torch.Tensor, cols: Optional[tuple] = None) -> torch.Tensor:
        """Extension (streaming): ``(B, n, 2K)`` [real | imag] features
        WITHOUT the per-signal z-score, for use with running moments.  ``cols=(col0, ncols)``
        computes only those frame centres."""
        X = self._as_f32(X, frames=True)
        if X.ndim == 1:
            X = X.unsqueeze(0)
        return self._run(X, _lib.MODE_STACK_UNNORM, cols=cols)

    # ------------------------------------------------------------------ reference helper kept for its contract
    def _truncate_frequencies(self, s: torch.Tensor, f: torch.Tensor):
        """Same contract as the reference helper (synchrosqueeze.py:91-111): raises ``ValueError``
        when ``truncate_freq`` is unset; otherwise slices the rows inside the band."""
        if not self.truncate_freq:
            raise ValueError(f"truncate_freq must be set, got: {self.truncate_freq}")
        klo, K = self.band()
        return s[klo:klo + K, :], f.reshape(-1)[klo:klo + K]
