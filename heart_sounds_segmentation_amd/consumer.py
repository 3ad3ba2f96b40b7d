"""Build-side counterpart of the feature CONSUMER, for end-to-end checks only (BASELINE config 4).

The reference's ``HeartSoundSegmenter`` (/root/reference/hss/model/segmenter.py:5-87) stays "as-is on
PyTorch-ROCm" (north_star) and is out of scope as code to accelerate; but nothing under
/root/reference exists on the GPU box, so the harness needs a module that loads the same
``state_dict`` (keys ``lstm_1.*``, ``lstm_2.*``, ``linear.*``) and computes the same function:
BiLSTM(in -> 2xH) -> ReLU -> Dropout(0.2) -> BiLSTM(2H -> 2xH), seeded with the first layer's final
(h, c) -> ReLU -> Dropout -> Linear(2H -> 4) -> LogSoftmax over classes (segmenter.py:70-87).
Stock ``nn.LSTM`` (MIOpen): plumbing, not the product.  Pinned by tests/golden/segmenter.npz, which
was produced by the reference class itself (tests/golden/make_golden.py).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn


class SegmenterHead(nn.Module):
    def __init__(self, input_size: int = 44, hidden_size: int = 240, batch_size: int = 50,
                 h0: Optional[torch.Tensor] = None, c0: Optional[torch.Tensor] = None):
        super().__init__()
        mk = dict(hidden_size=hidden_size, bidirectional=True, batch_first=True)
        self.lstm_1 = nn.LSTM(input_size=input_size, **mk)
        self.lstm_2 = nn.LSTM(input_size=2 * hidden_size, **mk)
        self.linear = nn.Linear(2 * hidden_size, 4)
        self.drop = nn.Dropout(0.2)
        # the reference keeps random, non-persistent initial states of shape (2, batch, H)
        # (segmenter.py:38-41), which ties the model to one batch size; they are inputs here
        shape = (2, batch_size, hidden_size)
        self.register_buffer("h0", h0 if h0 is not None else torch.randn(shape), persistent=False)
        self.register_buffer("c0", c0 if c0 is not None else torch.randn(shape), persistent=False)

    @classmethod
    def seeded_like_reference(cls, seed: int, input_size: int = 44, hidden_size: int = 240,
                              batch_size: int = 50) -> "SegmenterHead":
        """The module the reference constructor would build under ``torch.manual_seed(seed)``: it draws h0, c0
        first, then initialises lstm_1, lstm_2 and linear (segmenter.py:38-67); same draws, same order here, so a
        fixture only needs the seed and a checksum of the weights (tests/golden/segmenter_c4.npz)."""
        torch.manual_seed(seed)
        shape = (2, batch_size, hidden_size)
        h0, c0 = torch.randn(shape), torch.randn(shape)
        return cls(input_size, hidden_size, batch_size, h0=h0, c0=c0)

    def checksum(self) -> bytes:
        """SHA-256 over the state_dict (sorted keys) and h0 / c0, as written by tests/golden/make_golden.py."""
        import hashlib
        h = hashlib.sha256()
        sd = self.state_dict()
        for k in sorted(sd.keys()):
            h.update(k.encode())
            h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
        h.update(self.h0.detach().cpu().numpy().tobytes())
        h.update(self.c0.detach().cpu().numpy().tobytes())
        return h.digest()

    def forward(self, feats: torch.Tensor) -> torch.Tensor:
        y, carry = self.lstm_1(feats, (self.h0, self.c0))
        y, _ = self.lstm_2(self.drop(torch.relu(y)), carry)
        return torch.log_softmax(self.linear(self.drop(torch.relu(y))), dim=2)


def segment(fsst, head: SegmenterHead, windows: torch.Tensor) -> torch.Tensor:
    """windows (B, n) -> HIP FSST features (B, n, 2K), kept on the device -> (B, n, 4) log-probs."""
    return head(fsst.batch(windows))
