"""Multi-GPU sharding of the FSST path: windows are independent, so rank r
of G computes a contiguous block of the window index with NO data-path collective; one optional
RCCL all-gather (torch.distributed backend "nccl" == RCCL over xGMI) reassembles the feature batch
on every rank for the consumer (the BiLSTM).  One process per GPU.

The reference has no distributed code at all; this module is the
build's single parallelism strategy.  ``compute`` is injected so that the sharding / gather logic
is testable on CPU with the gloo backend (tests/test_dist.py).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block split: the first ``total % world`` ranks get one extra window."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError(f"shard_bounds: bad arguments total={total} world={world} rank={rank}")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sharded_features(compute: Callable[[torch.Tensor], torch.Tensor], X: torch.Tensor,
                     gather: bool = True, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """``X``: the FULL ``(B, n)`` batch, identical on every rank (e.g. read from shared storage).
    Each rank transforms only its block ``X[lo:hi]``.  ``gather=False`` returns the local block
    (features stay sharded, e.g. for data-parallel training); ``gather=True`` returns the full
    ``(B, ...)`` feature batch on every rank via ONE all-gather (ragged tails padded to the
    largest block, then trimmed)."""
    if not (dist.is_available() and dist.is_initialized()):
        return compute(X)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = X.shape[0]
    lo, hi = shard_bounds(B, world, rank)
    local = compute(X[lo:hi])
    if not gather or world == 1:
        return local
    return all_gather_blocks(local, B, group)


def all_gather_blocks(local: torch.Tensor, total: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """All-gather per-rank feature blocks (block split of ``total`` by ``shard_bounds``) into the
    full batch, in rank order.  Equal blocks: a single ``all_gather_into_tensor`` straight into the
    result.  Ragged: pad to the largest block, gather once, trim."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_bounds(total, world, r) for r in range(world)]
    counts = [h - l for l, h in sizes]
    if local.shape[0] != counts[rank]:
        raise ValueError(f"all_gather_blocks: local block has {local.shape[0]} rows, expected {counts[rank]}")
    return _gather_counts(local, counts, group)


def all_gather_ragged(local: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """All-gather blocks whose row counts are only known locally (recording-level corpus split: the number of
    windows per rank depends on the recording lengths): one tiny all-gather of the counts, then the payload."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    mine = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    allc = torch.empty(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(allc, mine, group=group)
    return _gather_counts(local, [int(c) for c in allc.tolist()], group)


def _gather_counts(local: torch.Tensor, counts, group) -> torch.Tensor:
    world, total = len(counts), sum(counts)
    tail = tuple(local.shape[1:])
    local = local.contiguous()
    if len(set(counts)) == 1:
        out = torch.empty((total,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    mx = max(counts)
    padded = torch.zeros((mx,) + tail, dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    buf = torch.empty((world * mx,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    return torch.cat([buf[r * mx: r * mx + counts[r]] for r in range(world)], dim=0)
