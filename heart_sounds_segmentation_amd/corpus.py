"""Batched counterpart of the reference's in-memory dataset loop:
``DavidSpringerHSS.__init__`` (/root/reference/hss/datasets/heart_sounds.py:155-169) turns every
recording into frames (``frame_signal``, stride 1000, length 2000) and calls the transform once per
frame on the CPU.  Here recordings are uploaded ONCE, in groups laid back to back in one buffer, and all frames of a
group are transformed by one call (``hssfsst_exec_list``: no 2x duplicated H2D copy of the overlapping frames; a single
recording's frames can also be read in place with ``hssfsst_exec_frames`` via ``FSST.batch(frame_batch(x))``).

Semantics kept (pinned by tests/golden/frame_signal.npz): recordings shorter than ``frame_len`` are
skipped (heart_sounds.py:161-162); ``L = floor((T - n)/stride)`` frames -- one fewer than fit --
(preprocess.py:40,48-52); labels are shifted to 0-based (``y - 1``, heart_sounds.py:164) and framed
identically; each item is ``(features (n, 2K) float32, labels (n,) int64)`` (heart_sounds.py:168).

How it is fed (round 3; round 2 moved 77.7 k windows/s from host recordings against 3.98 M/s device-resident):
  * ONE feature arena ``(frames, n, 2K)`` per call -- on the device, or in (pinned) host memory when the caller wants
    host tensors -- and ONE label arena; the returned ``FrameItems`` hands out views of them on demand (the reference's
    list of 26 136 tuples is built only if somebody asks for ``list(items)``);
  * the recordings of a group are packed into one of two PINNED staging buffers and uploaded on a side stream while
    the previous group is transformed (``hssfsst_exec_list`` writes straight into the group's rows of the arena);
  * host-returned features leave the device group by group on a third stream, overlapping the next group's transform.

Multi-GPU (BASELINE config C3): ``rank`` / ``world`` split the RECORDINGS in contiguous
blocks (``dist.shard_bounds``), so framing stays local to a rank and concatenating the ranks' item lists in rank
order is the single-process list; ``gather_features`` reassembles the feature tensor on every rank with one
(ragged) RCCL all-gather on the process group's device; a rank without frames takes part with an empty block.
"""
from __future__ import annotations

from collections.abc import Sequence as _SequenceABC
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import dist as hdist
from .framing import frame_batch, frame_starts

Item = Tuple[torch.Tensor, Optional[torch.Tensor]]


class FrameItems(_SequenceABC):
    """The reference dataset's ``self.data`` -- a sequence of ``(features (n, 2K), labels (n,) or None)`` -- as views of
    one feature arena and one label arena (no per-item allocation; ``items[i]``, ``len``, iteration and slicing work as
    on the reference's list)."""

    def __init__(self, features: torch.Tensor, labels: Optional[torch.Tensor]):
        self.features = features                          # (frames, n, 2K) float32, device or host
        self.labels = labels                              # (frames, n) int64 on the host, or None

    def __len__(self) -> int:
        return int(self.features.shape[0])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self.features[i], (self.labels[i] if self.labels is not None else None)


class CorpusBuilder:
    """The builder as an object: keeps its pinned staging buffers, device staging, streams and (when asked) the feature
    arena between calls, so that a second corpus of the same size pays for no allocation (page-locking the host
    buffers and hipMalloc of gigabytes cost more than the transform itself)."""

    def __init__(self, fsst, stride: int = 1000, frame_len: int = 2000, device: Optional[torch.device] = None,
                 windows_per_launch: int = 4096, pin_host: bool = True):
        self.fsst, self.stride, self.frame_len = fsst, int(stride), int(frame_len)
        self.dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.wpl, self.pin_host = int(windows_per_launch), bool(pin_host)
        self._cap_samples = self._cap_frames = self._cap_ring = 0
        self._bufs = None

    def _ensure(self, max_samples: int, max_frames: int, C: int, need_ring: bool) -> None:
        dev = self.dev
        if self._bufs is None:
            self._bufs = {"up": torch.cuda.Stream(dev), "down": torch.cuda.Stream(dev),
                          "up_done": [torch.cuda.Event() for _ in range(2)], "used": [torch.cuda.Event() for _ in range(2)],
                          "ring_free": [torch.cuda.Event() for _ in range(2)]}
        b = self._bufs
        if max_samples > self._cap_samples:
            b["stage_h"] = [torch.empty(max_samples, dtype=torch.float32, pin_memory=True) for _ in range(2)]
            b["stage_d"] = [torch.empty(max_samples, dtype=torch.float32, device=dev) for _ in range(2)]
            self._cap_samples = max_samples
        if max_frames > self._cap_frames:
            b["start_h"] = [torch.empty(max_frames, dtype=torch.int64, pin_memory=True) for _ in range(2)]
            b["start_d"] = [torch.empty(max_frames, dtype=torch.int64, device=dev) for _ in range(2)]
            self._cap_frames = max_frames
        if need_ring and max_frames * C > self._cap_ring:
            b["ring_d"] = [torch.empty((max_frames, self.frame_len, C), dtype=torch.float32, device=dev) for _ in range(2)]
            self._cap_ring = max_frames * C

    def build(self, recordings: Iterable[Tuple[torch.Tensor, Optional[torch.Tensor]]], keep_on_device: bool = False,
              rank: Optional[int] = None, world: Optional[int] = None, out: Optional[torch.Tensor] = None) -> FrameItems:
        """See ``build_features``.  ``out``: a feature arena of a previous call to write into (same shape, device arena
        for ``keep_on_device`` else host)."""
        fsst, stride, frame_len, dev = self.fsst, self.stride, self.frame_len, self.dev
        recs: Sequence = recordings if isinstance(recordings, (list, tuple)) else list(recordings)
        # argument errors are properties of the WHOLE call and are raised before the list is cut to this rank's shard: a rank that
        # raised alone left the others waiting in the gather that follows
        if dev.type == "cuda" and not (getattr(fsst, "stack", False) or getattr(fsst, "abs", False)):
            # (the raw transform is complex64 (frames, K, n), frequency-major: not the time-major float32 arena this builder fills)
            raise ValueError("CorpusBuilder.build: the transform must have stack=True or abs=True (time-major float32 features); "
                             "for the raw complex transform call FSST.frames per group of recordings")
        kept_all = [y is not None for x, y in recs if x.shape[0] >= frame_len]
        if 0 < sum(kept_all) < len(kept_all):
            raise ValueError("CorpusBuilder.build: some recordings carry labels and some do not; pass labels for all or for none")
        if world is not None and world > 1:
            lo, hi = hdist.shard_bounds(len(recs), int(world), int(rank or 0))
            recs = recs[lo:hi]
        recs = [(x.reshape(-1), y) for x, y in recs if x.shape[0] >= frame_len]       # heart_sounds.py:161-162
        lens_np = np.asarray([int(x.shape[0]) for x, _ in recs], dtype=np.int64)
        L_np = (lens_np - frame_len) // stride                                           # frame_signal: one fewer than fit
        nfr_np = np.where(L_np <= 0, 1, L_np).astype(np.int64)
        nfr = [int(v) for v in nfr_np]
        total = int(nfr_np.sum())
        # (decided on the WHOLE list, before the cut: the same answer on every rank -- a rank whose shard holds no frames returns an
        #  empty (0, frame_len) int64 tensor, not None, when the corpus is labelled, so that a gather of the labels finds every rank in it)
        have_labels = len(kept_all) > 0 and all(kept_all)
        # groups of about `windows_per_launch` frames (a launch per recording -- 33 frames -- leaves the chip idle)
        groups: List[Tuple[int, int]] = []
        g0, acc = 0, 0
        for i, k in enumerate(nfr):
            acc += k
            if acc >= self.wpl:
                groups.append((g0, i + 1))
                g0, acc = i + 1, 0
        if g0 < len(recs):
            groups.append((g0, len(recs)))
        labels = torch.empty((total, frame_len), dtype=torch.int64) if have_labels else None

        def fill_labels(a: int, b: int, row: int) -> None:
            for i in range(a, b):
                labels[row:row + nfr[i]] = frame_batch(recs[i][1] - 1, stride, frame_len)
                row += nfr[i]

        if dev.type != "cuda":
            # host-only stand-in transforms (tests of the grouping / framing logic): one synchronous call per group
            feats = None
            row = 0
            for a, b in groups:
                xs = [recs[i][0].to(torch.float32) for i in range(a, b)]
                st, base = [], 0
                for i, x in zip(range(a, b), xs):
                    st.append(torch.from_numpy(frame_starts(int(x.shape[0]), stride, frame_len)[0]) + base)
                    base += int(x.shape[0])
                blk = fsst.frames(torch.cat(xs), torch.cat(st), frame_len)
                if feats is None:
                    feats = torch.empty((total,) + tuple(blk.shape[1:]), dtype=blk.dtype)
                feats[row:row + blk.shape[0]] = blk
                if labels is not None:
                    fill_labels(a, b, row)
                row += int(blk.shape[0])
            if feats is None:
                feats = torch.empty((0, frame_len, 0), dtype=torch.float32)
            return FrameItems(feats, labels)

        plan = fsst._plan(fsst._device_index(torch.empty(0, device=dev)))
        C = plan.ofps
        shape = (total, frame_len, C)
        if out is not None and (tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous()
                                or out.is_cuda != bool(keep_on_device)):
            raise ValueError(f"CorpusBuilder.build: out must be a contiguous float32 {shape} arena "
                             f"{'on the device' if keep_on_device else 'in host memory'}")
        feats = out
        if feats is None:
            if keep_on_device:
                feats = torch.empty(shape, dtype=torch.float32, device=dev)
            else:
                try:
                    feats = torch.empty(shape, dtype=torch.float32, pin_memory=bool(self.pin_host and total > 0))
                except RuntimeError:                      # page-locking that much memory can be refused: pageable then
                    feats = torch.empty(shape, dtype=torch.float32)
        if total == 0:
            return FrameItems(feats, labels)
        max_samples = max(sum(int(recs[i][0].shape[0]) for i in range(a, b)) for a, b in groups)
        max_frames = max(sum(nfr[a:b]) for a, b in groups)
        self._ensure(max_samples, max_frames, C, not keep_on_device)
        B = self._bufs
        main, up, down = torch.cuda.current_stream(dev), B["up"], B["down"]
        stage_h, stage_d, start_h, start_d = B["stage_h"], B["stage_d"], B["start_h"], B["start_d"]
        up_done, used, ring_free = B["up_done"], B["used"], B["ring_free"]
        ring_d = B.get("ring_d")
        up.wait_stream(main)                             # (the staging buffers may still be read by an earlier call's work)

        # the host side of a group is ONE native call (hssfsst_pack_recordings: threaded copies into the pinned staging
        # buffer + the frame starts): per recording Python does nothing but hand over a pointer.  (Per-recording copy_ /
        # numpy calls were 0.29 ms per recording: 115 k windows/s however fast the device is.)
        from . import _lib
        import ctypes
        held = [x if (x.dtype == torch.float32 and x.is_contiguous() and not x.is_cuda) else x.detach().to("cpu", torch.float32).contiguous()
                for x, _ in recs]
        ptrs_np = np.asarray([x.data_ptr() for x in held], dtype=np.uint64)
        pos_np = np.concatenate([[0], np.cumsum(lens_np)])
        fr_np = np.concatenate([[0], np.cumsum(nfr_np)])
        L = _lib.lib()

        def pack(gi: int) -> Tuple[int, int]:
            """Host side of group gi: recordings back to back into pinned staging, frame starts; upload on `up`."""
            a, b = groups[gi]
            buf = gi & 1
            if gi >= 2:
                used[buf].synchronize()                  # the transform of group gi - 2 no longer reads this staging pair
            pos, nf = int(pos_np[b] - pos_np[a]), int(fr_np[b] - fr_np[a])
            got = L.hssfsst_pack_recordings(ctypes.c_void_p(ptrs_np[a:b].ctypes.data), ctypes.c_void_p(lens_np[a:b].ctypes.data), b - a,
                                            stride, frame_len, ctypes.c_void_p(stage_h[buf].data_ptr()), int(stage_h[buf].numel()),
                                            ctypes.c_void_p(start_h[buf].data_ptr()), int(start_h[buf].numel()), 0)
            if got != nf:
                _lib.check(int(got) if got < 0 else _lib.E_INVAL, "hssfsst_pack_recordings")
            with torch.cuda.stream(up):
                stage_d[buf][:pos].copy_(stage_h[buf][:pos], non_blocking=True)
                start_d[buf][:nf].copy_(start_h[buf][:nf], non_blocking=True)
                up_done[buf].record(up)
            return pos, nf

        row = 0
        sizes = pack(0)
        for gi in range(len(groups)):
            buf = gi & 1
            pos, nf = sizes
            main.wait_event(up_done[buf])
            if keep_on_device:
                dst = feats[row:row + nf]
            else:
                if gi >= 2:
                    main.wait_event(ring_free[buf])
                dst = ring_d[buf][:nf]
            fsst.frames(stage_d[buf][:pos], start_d[buf][:nf], frame_len, out=dst)
            used[buf].record(main)
            if not keep_on_device:
                down.wait_stream(main)
                with torch.cuda.stream(down):
                    feats[row:row + nf].copy_(dst, non_blocking=True)
                    ring_free[buf].record(down)
            if gi + 1 < len(groups):
                sizes = pack(gi + 1)                     # host packing + upload of the next group overlap this transform
            if labels is not None:                       # (host work, also overlapped)
                fill_labels(groups[gi][0], groups[gi][1], row)
            row += nf
        if not keep_on_device:
            down.synchronize()
        main.synchronize()
        fsst.check()
        return FrameItems(feats, labels)


def build_features(recordings: Iterable[Tuple[torch.Tensor, Optional[torch.Tensor]]], fsst,
                   stride: int = 1000, frame_len: int = 2000, device: Optional[torch.device] = None,
                   keep_on_device: bool = False, rank: Optional[int] = None,
                   world: Optional[int] = None, windows_per_launch: int = 4096, pin_host: bool = True) -> FrameItems:
    """``recordings``: iterable of ``(x (T,) float32, y (T,) int64 labels in 1..4 or None)``.
    Returns what the reference dataset would hold in ``self.data`` (``in_memory=True, framing=True``);
    with ``world`` > 1 only the part of it that comes from this rank's block of recordings.
    ``keep_on_device=True`` leaves the features on the GPU (a GPU consumer follows: BASELINE config C4);
    otherwise they are returned in host memory (pinned when ``pin_host``), as the reference's CPU tensors.
    (One-shot form of ``CorpusBuilder``, which keeps its staging buffers between calls.)"""
    return CorpusBuilder(fsst, stride, frame_len, device, windows_per_launch, pin_host).build(
        recordings, keep_on_device=keep_on_device, rank=rank, world=world)


def gather_features(items, group=None, out_device: Optional[torch.device] = None) -> torch.Tensor:
    """All-gather this rank's features (rank order == recording order) into the full ``(windows, n, 2K)`` tensor on
    every rank.  ``items``: a ``FrameItems`` (its arena is the send block: no stacking) or a list of ``(features,
    labels)``; an empty one is fine as long as some rank has frames.  Under RCCL the exchange -- and the result, unless
    ``out_device`` says otherwise -- lives on this process's GPU.  Without an initialised process group: the block."""
    if isinstance(items, FrameItems):
        local = items.features
    elif len(items) > 0:
        local = torch.stack([f for f, _ in items], dim=0)
    else:
        local = None
    if local is not None and local.shape[0] == 0 and local.ndim < 3:
        local = None
    return hdist.all_gather_ragged(local, group, out_device=out_device)
