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


def _ragged_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = 3 if rank == 0 else 5                      # counts known only locally
    local = torch.arange(rows * 4, dtype=torch.float32).reshape(rows, 4) + 100.0 * rank
    full = hdist.all_gather_ragged(local)
    np.save(os.path.join(tmp, f"ragged{rank}.npy"), full.numpy())
    dist.destroy_process_group()


def test_all_gather_ragged_world2(tmp_path):
    mp.spawn(_ragged_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "ragged0.npy"), np.load(tmp_path / "ragged1.npy")
    want = np.concatenate([np.arange(12, dtype=np.float32).reshape(3, 4),
                           np.arange(20, dtype=np.float32).reshape(5, 4) + 100.0])
    assert np.array_equal(a, want) and np.array_equal(b, want)


def _empty_rank_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from heart_sounds_segmentation_amd import corpus
    # rank 1 holds nothing (e.g. its block of recordings is all shorter than a frame): it must still take part
    local = (torch.arange(3 * 2 * 5, dtype=torch.float32).reshape(3, 2, 5) if rank == 0 else None)
    full = hdist.all_gather_ragged(local)
    items = corpus.FrameItems(local if local is not None else torch.empty((0, 2, 5)), None)
    full2 = corpus.gather_features(items)
    np.save(os.path.join(tmp, f"empty{rank}.npy"), torch.stack([full, full2]).numpy())
    dist.destroy_process_group()


def test_all_gather_ragged_with_an_empty_rank(tmp_path):
    """A rank without rows (world > usable recordings, or a block of too-short recordings) joins the collectives with a
    0-row block and learns shape and dtype from the others (round 2 raised on that rank while its peers hung)."""
    mp.spawn(_empty_rank_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    want = np.arange(30, dtype=np.float32).reshape(3, 2, 5)
    for r in range(2):
        got = np.load(tmp_path / f"empty{r}.npy")
        assert np.array_equal(got[0], want) and np.array_equal(got[1], want)


# ---------------------------------------------------------------------------------------------- GPU
# The N > 1 path with the HIP kernels as the per-rank compute.  A gpurun box has ONE GPU and RCCL refuses two
# ranks on one device, so both ranks drive cuda:0 and the exchange runs over gloo (features staged to the host);
# the sharding, the recording-level corpus split and the reassembly are the product code paths.

def _corpus(T_list, seed):
    recs = []
    for i, T in enumerate(T_list):
        x = torch.from_numpy(synth.recording(T, seed=seed + i))
        y = torch.from_numpy(np.random.default_rng(seed + 100 + i).integers(1, 5, size=T).astype(np.int64))
        recs.append((x, y))
    return recs


def _gpu_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from heart_sounds_segmentation_amd import FSST, corpus
    torch.cuda.set_device(0)
    tf = FSST(1000, synth.kaiser_window(128, 0.5), truncate_freq=(25, 200), stack=True, device="cuda:0")
    X = torch.from_numpy(synth.pcg_windows(7, 2000, seed=11))            # ragged split: 4 + 3

    def compute(xb):
        return tf.batch(xb.cuda()).cpu()

    full = hdist.sharded_features(compute, X, gather=True)
    np.save(os.path.join(tmp, f"c2_{rank}.npy"), full.numpy())
    # C3: recording-level split of a small corpus stand-in, then the ragged reassembly
    recs = _corpus([5200, 1500, 7300, 4100, 9000], seed=21)              # one recording is too short and skipped
    items = corpus.build_features(recs, tf, rank=rank, world=world)
    feats = corpus.gather_features(items)
    np.save(os.path.join(tmp, f"c3_{rank}.npy"), feats.numpy())
    np.save(os.path.join(tmp, f"c3n_{rank}.npy"), np.asarray([len(items)]))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_hip_compute_world2(tmp_path):
    from heart_sounds_segmentation_amd import FSST, corpus
    mp.spawn(_gpu_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    tf = FSST(1000, synth.kaiser_window(128, 0.5), truncate_freq=(25, 200), stack=True, device="cuda:0")
    X = torch.from_numpy(synth.pcg_windows(7, 2000, seed=11))
    single = tf.batch(X.cuda()).cpu().numpy()
    for r in range(2):                                                    # bit-for-bit: the result does not depend on the split
        assert np.array_equal(np.load(tmp_path / f"c2_{r}.npy"), single)
    recs = _corpus([5200, 1500, 7300, 4100, 9000], seed=21)
    items = corpus.build_features(recs, tf)
    want = torch.stack([f for f, _ in items]).numpy()
    assert want.shape[0] == 3 + 5 + 2 + 7                                 # floor((T - 2000) / 1000) frames per recording
    n0, n1 = int(np.load(tmp_path / "c3n_0.npy")[0]), int(np.load(tmp_path / "c3n_1.npy")[0])
    assert n0 + n1 == want.shape[0] and n0 > 0 and n1 > 0
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"c3_{r}.npy"), want)


def _rccl_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    from heart_sounds_segmentation_amd import FSST, corpus
    tf = FSST(1000, synth.kaiser_window(128, 0.5), truncate_freq=(25, 200), stack=True, device="cuda:0")
    recs = _corpus([5200, 1500, 7300, 4100], seed=33)
    out = {}
    for keep in (False, True):                                            # host-returned (the default) and device-kept features
        items = corpus.build_features(recs, tf, rank=rank, world=world, keep_on_device=keep, windows_per_launch=4)
        feats = corpus.gather_features(items)                             # RCCL: runs -- and stays -- on the GPU
        assert feats.is_cuda and feats.shape == (3 + 5 + 2, 2000, 44)
        out[keep] = feats.cpu()
        assert torch.equal(out[keep], items.features.cpu())
    assert torch.equal(out[False], out[True])
    blk = hdist.all_gather_blocks(out[True].cuda(), out[True].shape[0])   # equal-block path (in-place RCCL all-gather)
    assert torch.equal(blk.cpu(), out[True])
    host = corpus.gather_features(items, out_device=torch.device("cpu"))
    assert not host.is_cuda and torch.equal(host, out[True])
    np.save(os.path.join(tmp, "rccl.npy"), out[True].numpy())
    dist.destroy_process_group()


@pytest.mark.gpu
def test_corpus_gather_over_rccl_one_rank(tmp_path):
    """The C3 exchange on the backend north_star names: a 1-rank RCCL process group (what one gpurun box can host) with
    host-returned AND device-kept features -- round 2 handed CPU tensors to the RCCL collective and had only ever run
    over gloo.  Features equal the single-process builder's."""
    from heart_sounds_segmentation_amd import FSST, corpus
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    tf = FSST(1000, synth.kaiser_window(128, 0.5), truncate_freq=(25, 200), stack=True, device="cuda:0")
    items = corpus.build_features(_corpus([5200, 1500, 7300, 4100], seed=33), tf)
    assert np.array_equal(np.load(tmp_path / "rccl.npy"), items.features.numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_bench_n_ranks_dry_run_over_gloo(world):
    """`bench.py --gpus N --dry-run-gloo`: the WHOLE N-rank path of the round-end scaling run -- self-launch through
    torch.distributed.run on 127.0.0.1, window sharding, the timed C2 steps with barrier + max over ranks, the all-gather of the
    feature blocks, the C3 recording split with its ragged gather, one JSON line with n_gpus = N -- with every rank on the one
    leased GPU and the exchange over gloo (RCCL refuses two ranks per device): what is left for an 8-GPU box is RCCL itself."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--dry-run-gloo", "--steps", "3", "--warmup", "1",
           "--settle-steps", "0", "--batch", "32", "--no-cpu-baseline", "--c3-recordings", "24"]
    res = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == world and d["steps"] == 3 and d["unit"] == "windows/s" and d["value"] > 0
    assert d["config"]["windows_per_gpu"] == 32 and "dry_run" in d["config"]
    assert "error" not in d.get("allgather", {}) and d["allgather"]["bytes_per_rank"] == 32 * 2000 * 44 * 4
    c3 = d["c3"]
    assert "error" not in c3, c3
    if "n_gpus" in c3:
        assert c3["n_gpus"] == world
    assert c3["allgather"]["shape_ok"] is True and "error" not in c3["allgather"]


@pytest.mark.gpu
def test_c_abi_allgather_one_rank_rccl():
    """hssfsst_allgather: the torch-free form of the path's one exchange -- ncclAllGather through the library,
    RCCL loaded with dlopen.  One rank (the only world a one-GPU box has): the communicator is made here with RCCL's own C API,
    the gather runs out of place and in place, a bounded wait returns.  (More than one rank has never run: no multi-GPU node.)"""
    import ctypes
    import torch
    from heart_sounds_segmentation_amd import _lib
    try:
        R = ctypes.CDLL("librccl.so.1")
    except OSError:
        pytest.skip("no librccl.so.1")

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    R.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
    R.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    R.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    uid = UniqueId()
    assert R.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    assert R.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        L = _lib.lib()
        src = torch.arange(3 * 2000 * 44, dtype=torch.float32, device="cuda")
        dst = torch.zeros_like(src)
        st = torch.cuda.current_stream().cuda_stream
        rc = L.hssfsst_allgather(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), src.numel(), comm, ctypes.c_void_p(st), 5000)
        assert rc == 0, L.hssfsst_last_error()
        torch.cuda.synchronize()
        assert torch.equal(src, dst)
        dst.mul_(2.0)                                    # in place: the rank's block is its own slot of the result
        rc = L.hssfsst_allgather(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(dst.data_ptr()), dst.numel(), comm, ctypes.c_void_p(st), 0)
        assert rc == 0, L.hssfsst_last_error()
        torch.cuda.synchronize()
        assert torch.equal(dst, src * 2.0)
        assert L.hssfsst_allgather(None, ctypes.c_void_p(dst.data_ptr()), 4, comm, None, 0) != 0      # bad argument: status, no throw
    finally:
        R.ncclCommDestroy(comm)
