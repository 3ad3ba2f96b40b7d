#!/usr/bin/env python3
"""bench.py -- PCG windows/sec of the FSST feature path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path (synchrosqueeze core kernel + z-score kernel) over one batch of
synthetic input already resident in HBM.  Workload at every N: BASELINE.json configs[1] per GPU
("Batch of 1024 synthetic 2000-sample fp32 PCG windows", fs 1 kHz, Kaiser(128, 0.5), band
[25, 200] Hz, stack=True -> (1024, 2000, 44) fp32); with N > 1 ranks the windows are sharded
(weak scaling: 1024 per rank, no data-path collective); the optional RCCL all-gather that
reassembles the feature batch for the consumer is timed separately and reported beside `value`.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_WINDOW = 2000 * 4 + 2000 * 44 * 4      # 8 000 read + 352 000 written
HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: 8.0 TB/s spec


def pmc_traffic():
    """HBM bytes per core-kernel launch from the committed PMC pass of this same command
    (profiles/r01_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected as the
    MI355X guide prescribes).  None when the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as fh:
            return int(json.load(fh)["fsst_core128_kernel"]["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(X, w, budget_s=12.0):
    """The oracle ("port" of the reference CPU path: fp64 fsst + wrapper epilogue) on the host
    cores of this box, OpenMP over windows, on a bounded sample of the same workload."""
    import oracle
    oracle.build()
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, oracle.max_threads()))
    oracle.features(X[:threads], 1000, w, (25, 200), "stack", nthreads=threads)      # warm-up
    done, t0 = 0, time.perf_counter()
    chunk = max(threads * 4, 32)
    pos = 0
    while True:
        xs = X[pos:pos + chunk]
        if xs.shape[0] == 0:
            pos = 0
            continue
        oracle.features(xs, 1000, w, (25, 200), "stack", nthreads=threads)
        done += xs.shape[0]
        pos += chunk
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    return {"value": round(done / el, 2), "unit": "windows/s", "cores": threads, "kind": "port",
            "sample": f"{done} of the workload's 2000-sample windows in {el:.1f} s, fp64 C restatement "
                      f"(oracle/fsst_oracle.c), OpenMP over windows on {threads} of {cores} host cores"}


def self_launch(ngpus: int) -> int:
    """Re-exec this command under torch.distributed.run with `ngpus` ranks on this node; fails loudly when fewer
    GPUs are visible (never a silent 1-rank run that reports n_gpus = N)."""
    import socket
    import subprocess

    import torch
    have = torch.cuda.device_count()
    if have < ngpus:
        print(f"bench.py: --gpus {ngpus} requested but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--settle-steps", type=int, default=300,
                    help="untimed steps run BEFORE the W warm-up steps so the GPU reaches its sustained clocks "
                         "(a 20 ms burst from idle measures the clock ramp: core kernel 0.236 ms vs 0.205 ms sustained)")
    ap.add_argument("--batch", type=int, default=1024, help="windows per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    # N > 1 without a launcher: start the N ranks ourselves (one process per GPU, RCCL rendezvous on 127.0.0.1),
    # exactly as the driver would:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py ...
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    from heart_sounds_segmentation_amd import FSST, dist as hdist, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if torch.cuda.device_count() < min(world, int(os.environ.get("LOCAL_WORLD_SIZE", world))):
        raise SystemExit(f"bench.py: --gpus {args.gpus} needs {world} visible GPUs, found {torch.cuda.device_count()}")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("HSS_BENCH_FORCE_DIST") == "1"   # the latter: 1-rank RCCL smoke test
    torch.cuda.set_device(local)
    if use_dist:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local),
                                timeout=datetime.timedelta(seconds=300))
    dev = torch.device("cuda", local)

    n, B = 2000, args.batch
    w = synth.kaiser_window(128, 0.5)
    Xh = synth.pcg_windows(B, n, seed=synth.SEED + rank)
    X = torch.from_numpy(Xh).to(dev)
    tf = FSST(1000, w, truncate_freq=(25, 200), stack=True, device=dev)
    out = torch.empty((B, n, 44), dtype=torch.float32, device=dev)

    def sync_all():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.settle_steps, 0)):               # clock ramp from idle, untimed (see --settle-steps)
        tf.batch(X, out=out)
    for _ in range(args.warmup):
        tf.batch(X, out=out)
    sync_all()
    tf.set_timing(True, local)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tf.batch(X, out=out)
    sync_all()
    elapsed = time.perf_counter() - t0
    core_ms, norm_ms, ncalls = tf.timing(local)
    tf.set_timing(False, local)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # optional exchange step, timed on its own: all-gather of the per-rank feature blocks
    gather = None
    if use_dist:
        try:
            full = hdist.all_gather_blocks(out, B * world)          # warm-up (allocates, builds rings)
            sync_all()
            g0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                full = hdist.all_gather_blocks(out, B * world)
            sync_all()
            g = torch.tensor([(time.perf_counter() - g0) / reps], dtype=torch.float64, device=dev)
            dist.all_reduce(g, op=dist.ReduceOp.MAX)
            gms = float(g.item()) * 1e3
            step_ms = elapsed / args.steps * 1e3
            gather = {"allgather_ms": round(gms, 3), "bytes_per_rank": B * n * 44 * 4,
                      "value_with_allgather": round(B * world / ((step_ms + gms) * 1e-3), 1)}
            del full
        except Exception as e:                                   # never lose the bench line to the side measurement
            gather = {"error": f"{type(e).__name__}: {e}"[:200]}
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        core_avg_s = (core_ms / max(ncalls, 1)) * 1e-3
        achieved = BYTES_PER_WINDOW * B / core_avg_s / 1e9 if core_avg_s > 0 else 0.0
        line = {
            "metric": "PCG windows/sec FSST (1 kHz, 2000-sample)", "value": round(value, 1),
            "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C2: {B} x {n} fp32 synthetic PCG windows per GPU, fs=1000, "
                                   "Kaiser(128,0.5), band [25,200] Hz, stack=True -> (2000,44) fp32",
                       "windows_per_gpu": B, "clock_settle_steps": max(args.settle_steps, 0),
                       "parallelism": f"window-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": "fsst_core128_kernel<16, 8, 64, true, 16, 3>",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": pmc_traffic() if B == 1024 else None,
                         "algorithmic_bytes_per_launch": BYTES_PER_WINDOW * B,
                         "avg_launch_ms": round(core_ms / max(ncalls, 1), 4),
                         "normalize_avg_launch_ms": round(norm_ms / max(ncalls, 1), 4),
                         "launches_timed": ncalls},
        }
        if gather:
            line["allgather"] = gather
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(Xh, w, args.cpu_budget)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
