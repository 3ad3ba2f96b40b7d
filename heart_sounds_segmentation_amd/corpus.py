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

Multi-GPU (BASELINE config C3): ``rank`` / ``world`` split the RECORDINGS in contiguous
blocks (``dist.shard_bounds``), so framing stays local to a rank and concatenating the ranks' item lists in rank
order is the single-process list; ``gather_features`` reassembles the feature tensor on every rank with one
(ragged) RCCL all-gather.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch

from . import dist as hdist
from .framing import frame_batch, frame_starts

Item = Tuple[torch.Tensor, Optional[torch.Tensor]]


def build_features(recordings: Iterable[Tuple[torch.Tensor, Optional[torch.Tensor]]], fsst,
                   stride: int = 1000, frame_len: int = 2000, device: Optional[torch.device] = None,
                   keep_on_device: bool = False, rank: Optional[int] = None,
                   world: Optional[int] = None, windows_per_launch: int = 4096) -> List[Item]:
    """``recordings``: iterable of ``(x (T,) float32, y (T,) int64 labels in 1..4 or None)``.
    Returns the list the reference dataset would hold in ``self.data`` (``in_memory=True, framing=True``);
    with ``world`` > 1 only the part of it that comes from this rank's block of recordings.

    Recordings are taken in groups of about ``windows_per_launch`` frames: a group's recordings are laid back to back
    in ONE host buffer, uploaded once, and all their frames are transformed by one ``hssfsst_exec_list`` call
    (``FSST.frames``) -- a launch per recording (33 frames) leaves the chip idle most of the time."""
    items: List[Item] = []
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if world is not None and world > 1:
        recs: Sequence = recordings if isinstance(recordings, (list, tuple)) else list(recordings)
        lo, hi = hdist.shard_bounds(len(recs), int(world), int(rank or 0))
        recordings = recs[lo:hi]

    group: List[Tuple[torch.Tensor, Optional[torch.Tensor]]] = []
    gframes = 0

    def flush() -> None:
        nonlocal group, gframes
        if not group:
            return
        xs = [x.reshape(-1).to(torch.float32) for x, _ in group]
        starts, base = [], 0
        for x in xs:
            st, _ = frame_starts(int(x.shape[0]), stride, frame_len)
            starts.append(torch.from_numpy(st) + base)
            base += int(x.shape[0])
        xd = torch.cat(xs).to(dev)                                             # ONE upload of the group's recordings
        feats = fsst.frames(xd, torch.cat(starts), frame_len)                  # (sum L, n, 2K): one launch
        if not keep_on_device:
            feats = feats.cpu()
        rows = feats.unbind(0)                                                 # views of the group's feature block
        k = 0
        for (x, y), st in zip(group, starts):
            L = int(st.shape[0])
            if y is not None:
                labels = frame_batch((y - 1), stride, frame_len).contiguous().unbind(0)   # one copy per recording
                items.extend(zip(rows[k:k + L], labels))
            else:
                items.extend((r, None) for r in rows[k:k + L])
            k += L
        group, gframes = [], 0

    for x, y in recordings:
        if x.shape[0] < frame_len:
            continue
        group.append((x, y))
        gframes += len(frame_starts(int(x.reshape(-1).shape[0]), stride, frame_len)[0])
        if gframes >= windows_per_launch:
            flush()
    flush()
    return items


def gather_features(items: List[Item], group=None) -> torch.Tensor:
    """Stack this rank's features and all-gather them (rank order == recording order) into the full
    ``(windows, n, 2K)`` tensor on every rank.  Without an initialised process group: just the stack."""
    if not items:
        raise ValueError("gather_features: this rank holds no items (give every rank at least one recording)")
    local = torch.stack([f for f, _ in items], dim=0)
    return hdist.all_gather_ragged(local, group)
