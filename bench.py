#!/usr/bin/env python3
"""bench.py -- PCG windows/sec of the FSST feature path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path (synchrosqueeze transform + per-window z-score) over one batch of
synthetic input already resident in HBM.  Workload at every N: BASELINE.json configs[1] per GPU
("Batch of 1024 synthetic 2000-sample fp32 PCG windows", fs 1 kHz, Kaiser(128, 0.5), band
[25, 200] Hz, stack=True -> (1024, 2000, 44) fp32); with N > 1 ranks the windows are sharded
(weak scaling: 1024 per rank, no data-path collective); the optional RCCL all-gather that
reassembles the feature batch for the consumer is timed separately and reported beside `value`.

`python bench.py --gpus N` without a launcher starts the N ranks itself (torch.distributed.run on
127.0.0.1) and fails loudly when fewer GPUs are visible.  `--config c5` measures BASELINE config 5
(streaming, 64 channels x 4 kHz, 128 new samples per step) instead and prints its own JSON line.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_WINDOW = 2000 * 4 + 2000 * 44 * 4      # 8 000 read + 352 000 written
HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3                         # MI355X_MICROARCH.md: vector = matrix fp32 peak
MFMA_FLOP = 2048                                 # one v_mfma_f32_16x16x4_f32: 16 x 16 x 4 x 2
VALU_FLOP_PER_LANE = 2.0                         # estimate: the VALU mix is ~half packed FMA (4), ~half packed add/mul (2), rest 0-1


def csrc_digest():
    """SHA-256 of the device code of the benchmarked kernels (csrc/fsst_mfma128.hpp + csrc/fsst_kernels.hpp): PMC
    figures committed under profiles/ are only quoted for the kernels they were measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "heart_sounds_segmentation_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f in ("fsst_mfma128.hpp", "fsst_kernels.hpp"):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def pmc_profile(kernel_substr):
    """Per-launch PMC figures of the dominant kernel from the committed rocprofv3 passes of this same command
    (profiles/r02_pmc.json, written by tools/profile_round.sh: separate --pmc passes, FETCH_SIZE corrected as the
    MI355X guide prescribes).  None when the file is absent or was measured on other kernel sources."""
    path = os.path.join(ROOT, "profiles", "r02_pmc.json")
    try:
        with open(path) as fh:
            prof = json.load(fh)
        if prof.get("csrc_sha256") != csrc_digest():
            return None
        for name, v in prof["kernels"].items():
            if kernel_substr in name:
                return v
    except (OSError, KeyError, ValueError):
        pass
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


def bench_c5(args):
    """BASELINE config 5: 64 channels x 4 kHz, 128 new samples per channel and step, rolling FSST (nwin 512 = the
    same 128 ms window and 7.8125 Hz grid as the canonical configuration), running-moments z-score.  Reports
    device-resident steps/s and the HOST-VISIBLE latency of a step: last sample of a chunk in a host buffer ->
    its (64, 128, 44) features in a host buffer (pinned staging, H2D, kernels, D2H, one synchronisation)."""
    import numpy as np
    import torch
    from scipy.signal import get_window

    from heart_sounds_segmentation_amd import synth
    from heart_sounds_segmentation_amd.streaming import StreamingFSST
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ch, fs, chunk, nwin = 64, 4000, 128, 512
    w = get_window(("kaiser", 0.5), nwin, fftbins=False)
    st = StreamingFSST(ch, fs, w, truncate_freq=(25, 200), chunk=chunk, device=dev)
    steps, warm = args.steps, max(args.warmup, 20)
    xh = synth.pcg_windows(ch, chunk * 64, fs=fs, seed=2)
    xd = torch.from_numpy(xh).to(dev)
    for i in range(warm):
        st.step(xd[:, (i % 64) * chunk:(i % 64 + 1) * chunk])
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        st.step(xd[:, (i % 64) * chunk:(i % 64 + 1) * chunk])
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    lat = []
    nlat = min(steps, 500)
    for i in range(warm):
        st.step_host(xh[:, (i % 64) * chunk:(i % 64 + 1) * chunk])
    for i in range(nlat):
        xs = xh[:, (i % 64) * chunk:(i % 64 + 1) * chunk]
        t1 = time.perf_counter()
        st.step_host(xs)
        lat.append(time.perf_counter() - t1)
    lat = np.sort(np.asarray(lat)) * 1e3
    line = {"metric": "streaming FSST steps/sec (64 ch x 4 kHz, 128 new samples per step)", "value": round(1.0 / dt, 1),
            "unit": "steps/s", "n_gpus": 1, "steps": steps, "warmup": warm, "ms_per_step": round(dt * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C5: 64 channels x 4 kHz, chunk 128, Kaiser(512, 0.5), band [25,200] Hz, running-moments "
                                   "z-score -> (64, 128, 44) fp32 per step", "realtime_factor": round((chunk / fs) / dt, 1),
                       "lookahead_ms": round((nwin // 2 - 1) / fs * 1e3, 2)},
            "latency_host_visible_ms": {"median": round(float(np.median(lat)), 4), "p99": round(float(lat[int(0.99 * (len(lat) - 1))]), 4),
                                        "min": round(float(lat[0]), 4), "samples": int(len(lat)),
                                        "path": "pinned host chunk (32 KiB) -> H2D -> kernels -> D2H (1.44 MB) -> stream sync"},
            "roofline": {"bound": "hbm", "achieved": round((ch * chunk * 4 + ch * chunk * 44 * 4) / dt / 1e9, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round((ch * chunk * 4 + ch * chunk * 44 * 4) / dt / 1e9 / HBM_PEAK_GBS, 6), "traffic": None,
                         "note": "launch-latency bound: 1.47 MB per step"}}
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--settle-steps", type=int, default=300,
                    help="extra untimed steps run BEFORE the W warm-up steps so the GPU reaches its sustained clocks "
                         "(a 20 ms burst from idle measures the clock ramp); counted in the reported `warmup`")
    ap.add_argument("--batch", type=int, default=1024, help="windows per GPU")
    ap.add_argument("--config", choices=("c2", "c5"), default="c2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.config == "c5":
        if args.gpus != 1:
            ap.error("--config c5 is a single-GPU measurement")
        return bench_c5(args)

    # N > 1 without a launcher: start the N ranks ourselves (one process per GPU, RCCL rendezvous on 127.0.0.1),
    # exactly as the driver would:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py ...
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

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
        world = dist.get_world_size()                      # what RCCL actually sees
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

    untimed = max(args.settle_steps, 0) + args.warmup        # all of them are warm-up steps and are reported as such
    for _ in range(untimed):
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
    fused = tf.check(local)                                   # raises if a kernel reported a failed internal wait
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
        dom_ms = core_ms / max(ncalls, 1)                        # the dominant kernel: fused -> the whole transform + z-score
        alg = BYTES_PER_WINDOW * B
        achieved = alg / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        step_gbs = alg / (ms_per_step * 1e-3) / 1e9
        kname = ("fsst_core128_kernel<16, 8, 64, true, 16, 3, true>" if fused else "fsst_core128_kernel<16, 8, 64, true, 16, 3, false>")
        prof = pmc_profile(kname) if B == 1024 else None
        roof = {"bound": "hbm", "kernel": kname + (" (transform + z-score fused)" if fused else " (transform; z-score is a second kernel)"),
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": prof.get("hbm_bytes_per_launch") if prof else None,
                "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(dom_ms, 4),
                "other_kernels_avg_ms": round(norm_ms / max(ncalls, 1), 4), "launches_timed": ncalls,
                # the whole path (every kernel of a step + gaps), the figure north_star's 40 % is about
                "step_achieved": round(step_gbs, 2), "step_frac": round(step_gbs / HBM_PEAK_GBS, 5)}
        if prof and "SQ_INSTS_MFMA" in prof and "SQ_INSTS_VALU" in prof and dom_ms > 0:
            flop = prof["SQ_INSTS_MFMA"] * MFMA_FLOP + prof["SQ_INSTS_VALU"] * 64 * VALU_FLOP_PER_LANE
            roof["fp32_tflops"] = round(flop / (dom_ms * 1e-3) / 1e12, 2)
            roof["fp32_frac"] = round(flop / (dom_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)
            roof["fp32_note"] = (f"{int(prof['SQ_INSTS_MFMA'])} v_mfma_f32_16x16x4_f32 x {MFMA_FLOP} FLOP + "
                                 f"{int(prof['SQ_INSTS_VALU'])} VALU wave-instructions x 64 lanes x {VALU_FLOP_PER_LANE} FLOP (estimate) "
                                 f"per launch (profiles/r02_pmc.json); peak {FP32_PEAK_TFLOPS} TFLOP/s")
        line = {
            "metric": "PCG windows/sec FSST (1 kHz, 2000-sample)", "value": round(value, 1),
            "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": untimed,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C2: {B} x {n} fp32 synthetic PCG windows per GPU, fs=1000, "
                                   "Kaiser(128,0.5), band [25,200] Hz, stack=True -> (2000,44) fp32",
                       "windows_per_gpu": B, "warmup_requested": args.warmup, "clock_settle_steps": max(args.settle_steps, 0),
                       "parallelism": f"window-sharded x{world}, no data-path collective",
                       "zscore": "fused into the transform kernel" if fused else "second kernel"},
            "roofline": roof,
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
