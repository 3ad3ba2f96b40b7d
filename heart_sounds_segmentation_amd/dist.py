"""Multi-GPU sharding of the FSST path: windows are independent, so rank r
of G computes a contiguous block of the window index with NO data-path collective; one optional
RCCL all-gather (torch.distributed backend "nccl" == RCCL over xGMI) reassembles the feature batch
on every rank for the consumer (the BiLSTM).  One process per GPU.

The reference has no distributed code at all; this module is the
build's single parallelism strategy.  ``compute`` is injected so that the sharding / gather logic
is testable on CPU with the gloo backend (tests/test_dist.py).

Where the exchange runs: RCCL moves device memory only, so under the ``nccl`` backend every tensor handed to a
collective lives on this process's current GPU (host features are staged there first and the result stays there unless
``out_device`` says otherwise); under ``gloo`` the tensors stay where they are.  Ragged blocks (the recording-level
corpus split: a rank's window count depends on its recordings' lengths) are gathered by ONE ``all_gather_into_tensor`` of blocks
padded to the largest count plus a compaction pass -- under RCCL and gloo alike: it is the one collective every N-rank RCCL job
exercises, and this code has never run on more than one GPU (no multi-GPU node in rounds 1-5); a rank that holds no rows takes
part with a block of padding.  ``HSSFSST_DIST_UNEVEN=1`` opts into a single ``dist.all_gather`` over unequal per-rank views of the
result instead (no staging buffer; untested on hardware, never used for empty blocks).
"""
from __future__ import annotations

import os

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block split: the first ``total % world`` ranks get one extra window."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError(f"shard_bounds: bad arguments total={total} world={world} rank={rank}")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def comm_device(group: Optional[dist.ProcessGroup] = None, like: Optional[torch.Tensor] = None) -> torch.device:
    """The device collectives of ``group`` must run on: this process's current GPU for RCCL (``nccl``), otherwise the
    device of ``like`` (gloo moves host memory, and device memory through the host)."""
    backend = str(dist.get_backend(group)).lower()
    if "nccl" in backend:
        return torch.device("cuda", torch.cuda.current_device())
    return like.device if like is not None else torch.device("cpu")


def sharded_features(compute: Callable[[torch.Tensor], torch.Tensor], X: torch.Tensor,
                     gather: bool = True, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """``X``: the FULL ``(B, n)`` batch, identical on every rank (e.g. read from shared storage).
    Each rank transforms only its block ``X[lo:hi]``.  ``gather=False`` returns the local block
    (features stay sharded, e.g. for data-parallel training); ``gather=True`` returns the full
    ``(B, ...)`` feature batch on every rank via ONE all-gather."""
    if not (dist.is_available() and dist.is_initialized()):
        return compute(X)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = X.shape[0]
    lo, hi = shard_bounds(B, world, rank)
    local = compute(X[lo:hi])
    if not gather or world == 1:
        return local
    return all_gather_blocks(local, B, group)


def all_gather_blocks(local: torch.Tensor, total: int, group: Optional[dist.ProcessGroup] = None,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """All-gather per-rank feature blocks (block split of ``total`` by ``shard_bounds``) into the
    full batch, in rank order.  Equal blocks: a single ``all_gather_into_tensor`` straight into the
    result; ragged: ONE ``all_gather_into_tensor`` of blocks padded to the largest count + a compaction
    of the other ranks' rows (``_gather_counts``; the single uneven collective is opt-in,
    ``HSSFSST_DIST_UNEVEN=1``).  Never run under RCCL with more than one rank (no multi-GPU node)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [h - l for l, h in (shard_bounds(total, world, r) for r in range(world))]
    if local.shape[0] != counts[rank]:
        raise ValueError(f"all_gather_blocks: local block has {local.shape[0]} rows, expected {counts[rank]}")
    return _gather_counts(local, counts, tuple(local.shape[1:]), group, out)


_DTYPES = [torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.int64, torch.int32, torch.uint8, torch.bool,
           torch.complex64]


def all_gather_ragged(local: Optional[torch.Tensor], group: Optional[dist.ProcessGroup] = None,
                      tail: Optional[Sequence[int]] = None, dtype: Optional[torch.dtype] = None,
                      out_device: Optional[torch.device] = None) -> torch.Tensor:
    """All-gather blocks whose row counts are only known locally (recording-level corpus split): one tiny all-gather
    of {rows, trailing shape, dtype} per rank, then the payload straight into the result.  A rank without rows passes
    ``local=None`` (or a 0-row tensor): it learns the trailing shape and dtype from the others."""
    if not (dist.is_available() and dist.is_initialized()):
        if local is None:
            raise ValueError("all_gather_ragged: nothing to gather (no process group, no block)")
        return local if out_device is None else local.to(out_device)
    world = dist.get_world_size(group)
    if local is not None:
        tail, dtype = tuple(local.shape[1:]), local.dtype
    tail = tuple(int(t) for t in (tail or ()))
    if len(tail) > 6:
        raise ValueError("all_gather_ragged: at most 6 trailing dimensions")
    dev = comm_device(group, local)
    code = _DTYPES.index(dtype) if dtype in _DTYPES else -1
    rows = int(local.shape[0]) if local is not None else 0
    known = 1 if (local is not None or tail) and code >= 0 else 0
    mine = torch.tensor([rows, known, code, len(tail)] + list(tail) + [0] * (6 - len(tail)), dtype=torch.int64, device=dev)
    meta = torch.empty((world, mine.numel()), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(meta.view(-1), mine, group=group)
    meta_l: List[List[int]] = meta.cpu().tolist()
    counts = [int(m[0]) for m in meta_l]
    src = next((m for m in meta_l if m[1] == 1 and m[0] > 0), None) or next((m for m in meta_l if m[1] == 1), None)
    if src is None:
        raise ValueError("all_gather_ragged: no rank knows the block shape (every rank is empty and none passed `tail`)")
    tail, dtype = tuple(src[4:4 + src[3]]), _DTYPES[src[2]]
    # every rank holds all the metadata: a mismatch is raised on EVERY rank (raised only where it occurs, the others went on
    # into the collective and hung).  Blocks without rows carry no data and are not held to the shape.
    bad = [r for r, m in enumerate(meta_l) if m[0] > 0 and (m[1] != 1 or tuple(m[4:4 + m[3]]) != tail or m[2] != src[2])]
    if bad:
        raise ValueError(f"all_gather_ragged: the blocks of ranks {bad} do not match the trailing shape {tail} {dtype} of the others")
    if local is None or local.shape[0] == 0:
        local = torch.empty((0,) + tail, dtype=dtype, device=dev)
    full = _gather_counts(local, counts, tail, group, None)
    return full if out_device is None else full.to(out_device)


def _gather_counts(local: torch.Tensor, counts: Sequence[int], tail: Tuple[int, ...], group,
                   out: Optional[torch.Tensor]) -> torch.Tensor:
    world, rank, total = len(counts), dist.get_rank(group), sum(counts)
    dev = comm_device(group, local)
    if out is None:
        out = torch.empty((total,) + tuple(tail), dtype=local.dtype, device=dev)
    elif tuple(out.shape) != (total,) + tuple(tail) or out.device != dev or out.dtype != local.dtype or not out.is_contiguous():
        raise ValueError("gather: `out` must be a contiguous (total, ...) tensor of the block dtype on the communication device")
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    mine = out[offs[rank]: offs[rank + 1]]
    in_place = local.data_ptr() == mine.data_ptr() and local.device == dev      # (a block computed in place in `out` needs no copy)
    nccl = dev.type == "cuda" and "nccl" in str(dist.get_backend(group)).lower()
    if len(set(counts)) == 1:
        if not in_place:
            mine.copy_(local, non_blocking=True)
        # (RCCL gathers in place when the send buffer is the rank's own slot of the receive buffer; gloo gets a copy)
        dist.all_gather_into_tensor(out, mine if nccl else mine.clone(), group=group)
        return out
    if nccl and min(counts) > 0 and os.environ.get("HSSFSST_DIST_UNEVEN", "0") == "1":
        # OPT-IN (HSSFSST_DIST_UNEVEN=1): one dist.all_gather over per-rank views of the result.  torch lowers unequal blocks under
        # NCCL / RCCL to a coalesced group of per-rank broadcasts (ProcessGroupNCCL::allgather, as recalled -- not checked against this
        # build); it has NEVER run here with more than one rank (no multi-GPU node in rounds 1-6), and whether RCCL takes empty blocks is
        # unknown, so blocks without rows never come this way.  The default below uses only all_gather_into_tensor of equal blocks --
        # the collective every N-rank RCCL job exercises -- at the price of a padded staging buffer and one compaction pass.
        if not in_place:
            mine.copy_(local, non_blocking=True)
        views = [out[offs[r]: offs[r + 1]] for r in range(world)]
        dist.all_gather(views, mine, group=group)
        return out
    # ragged blocks (RCCL by default, gloo always -- gloo insists on equal blocks): ONE all_gather_into_tensor of blocks padded to the
    # largest count, then the valid rows of the OTHER ranks' blocks into place; a rank without rows takes part with a block of padding.
    # Nothing is zero-filled that is overwritten anyway: the staging buffer is uninitialised memory, only the padding rows of this
    # rank's send block are cleared (they travel), and the rank's own block goes local -> pad -> (its slot of `out` straight from local).
    cmax = max(counts)
    stage = torch.empty((world, cmax) + tuple(tail), dtype=local.dtype, device=dev)
    pad = torch.empty((cmax,) + tuple(tail), dtype=local.dtype, device=dev)
    pad[: counts[rank]].copy_(local, non_blocking=True)
    if counts[rank] < cmax:
        pad[counts[rank]:].zero_()
    if not in_place:
        mine.copy_(local, non_blocking=True)
    dist.all_gather_into_tensor(stage.view((world * cmax,) + tuple(tail)), pad, group=group)
    for r in range(world):
        if counts[r] and r != rank:
            out[offs[r]: offs[r + 1]].copy_(stage[r, : counts[r]])
    return out
