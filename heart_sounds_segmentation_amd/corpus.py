"""Batched counterpart of the reference's in-memory dataset loop:
``DavidSpringerHSS.__init__`` (/root/reference/hss/datasets/heart_sounds.py:155-169) turns every
recording into frames (``frame_signal``, stride 1000, length 2000) and calls the transform once per
frame on the CPU.  Here a recording's frames go to the GPU as ONE strided view and one launch.

Semantics kept (pinned by tests/golden/frame_signal.npz): recordings shorter than ``frame_len`` are
skipped (heart_sounds.py:161-162); ``L = floor((T - n)/stride)`` frames -- one fewer than fit --
(preprocess.py:40,48-52); labels are shifted to 0-based (``y - 1``, heart_sounds.py:164) and framed
identically; each item is ``(features (n, 2K) float32, labels (n,) int64)`` (heart_sounds.py:168).
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Tuple

import torch

from .framing import frame_batch


def build_features(recordings: Iterable[Tuple[torch.Tensor, Optional[torch.Tensor]]], fsst,
                   stride: int = 1000, frame_len: int = 2000, device: Optional[torch.device] = None,
                   keep_on_device: bool = False) -> List[Tuple[torch.Tensor, Optional[torch.Tensor]]]:
    """``recordings``: iterable of ``(x (T,) float32, y (T,) int64 labels in 1..4 or None)``.
    Returns the list the reference dataset would hold in ``self.data`` (``in_memory=True, framing=True``)."""
    items: List[Tuple[torch.Tensor, Optional[torch.Tensor]]] = []
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    for x, y in recordings:
        if x.shape[0] < frame_len:
            continue
        frames = frame_batch(x.to(torch.float32), stride, frame_len)          # (L, n) view
        feats = fsst.batch(frames.to(dev))                                   # (L, n, 2K) on the GPU
        if not keep_on_device:
            feats = feats.cpu()
        labels = frame_batch((y - 1), stride, frame_len) if y is not None else None
        for i in range(feats.shape[0]):
            items.append((feats[i], labels[i].clone() if labels is not None else None))
    return items
