"""world_size-2 gloo tests (CPU) of the N>1 path: block sharding + the all-gather reassembly of
heart_sounds_segmentation_amd.dist, with the oracle injected as the per-rank compute."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from heart_sounds_segmentation_amd import dist as hdist, synth


def test_shard_bounds_cover_everything():
    for total in (0, 1, 7, 8, 1024, 26136):
        for world in (1, 2, 3, 4, 8):
            spans = [hdist.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [h - l for l, h in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        hdist.shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    w = synth.kaiser_window(128, 0.5)
    X = torch.from_numpy(synth.noise_windows(B, 300, seed=4))

    def compute(xb):
        return torch.from_numpy(oracle.features(xb.numpy(), 1000, w, (25, 200), "stack"))

    full = hdist.sharded_features(compute, X, gather=True)
    local = hdist.sharded_features(compute, X, gather=False)
    lo, hi = hdist.shard_bounds(B, world, rank)
    ok = full.shape == (B, 300, 44) and local.shape[0] == hi - lo and torch.equal(full[lo:hi], local)
    np.save(os.path.join(tmp, f"full{rank}.npy"), full.numpy())
    np.save(os.path.join(tmp, f"ok{rank}.npy"), np.asarray([ok]))
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [6, 5])      # equal blocks and a ragged tail
def test_sharded_allgather_world2(tmp_path, oracle_mod, B):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, B, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "full0.npy")
    b = np.load(tmp_path / "full1.npy")
    assert np.load(tmp_path / "ok0.npy")[0] and np.load(tmp_path / "ok1.npy")[0]
    assert np.array_equal(a, b, equal_nan=True)
    w = synth.kaiser_window(128, 0.5)
    ref = oracle_mod.features(synth.noise_windows(B, 300, seed=4), 1000, w, (25, 200), "stack")
    assert np.array_equal(a, ref)
