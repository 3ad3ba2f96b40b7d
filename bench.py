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
MFMA_F16_FLOP = 16384                            # one v_mfma_f32_16x16x32_f16: 16 x 16 x 32 x 2
VALU_FLOP_PER_LANE = 2.0                         # estimate: the VALU mix is ~half packed FMA (4), ~half packed add/mul (2), rest 0-1


def csrc_digest():
    """SHA-256 of every source under csrc/ and of the C header: PMC figures committed under profiles/ are only quoted
    for the library they were measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "heart_sounds_segmentation_amd", "csrc")
    files = [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith((".hip", ".hpp", ".h"))]
    files.append(os.path.join(ROOT, "include", "hssfsst.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def pmc_profile(kernel_substr):
    """Per-launch PMC figures of the dominant kernel from the committed rocprofv3 passes of this same command
    (profiles/rNN_pmc.json, written by tools/profile_round.sh: separate --pmc passes, FETCH_SIZE corrected as the
    MI355X guide prescribes): the newest round's file whose SHA-256 of csrc/ matches the sources in the tree.  None when
    there is none -- never figures measured on other sources."""
    import glob
    digest = csrc_digest()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")), reverse=True):
        try:
            with open(path) as fh:
                prof = json.load(fh)
            if prof.get("csrc_sha256") != digest:
                continue
            for name, v in prof["kernels"].items():
                if kernel_substr in name:
                    return dict(v, _file=os.path.relpath(path, ROOT))
        except (OSError, KeyError, ValueError):
            pass
    return None


def census_max_rel_err():
    """Largest per-column error of the canonical kernels against the float64 oracle over a committed 2 M-column census
    (profiles/rNN_split_fold_census.txt, tools/split_fold_census.py): what the 22-bit operands of the fold amount to.
    A COMMITTED figure: (value, file, whether the file names the sources in the tree)."""
    import glob
    digest = csrc_digest()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_split_fold_census.txt")), reverse=True):
        try:
            val, sha = None, None
            with open(path) as fh:
                for ln in fh:
                    if ln.startswith("arithmetic_max_rel_err"):
                        val = float(ln.split()[1])
                    if ln.startswith("csrc_sha256"):
                        sha = ln.split()[1]
            if val is not None:
                return {"value": val, "source": os.path.relpath(path, ROOT), "same_sources": bool(sha == digest)}
        except (OSError, ValueError):
            pass
    return None


def box_spread():
    """The dominant kernel's spread over GPU leases of one source tree (profiles/rNN_box_spread.json: the newest round's committed
    A/B baselines, one row per lease).  A run of this script sees ONE box; this says where it may sit among the others."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_box_spread.json")), reverse=True):
        try:
            with open(path) as fh:
                d = json.load(fh)
            return {"per_exec_us": {"min": d["min"], "median": d["median"], "max": d["max"]}, "leases": d["leases"],
                    "frac_of_peak": {"min": round(BYTES_PER_WINDOW * 1024 / (d["max"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                     "max": round(BYTES_PER_WINDOW * 1024 / (d["min"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
                    "source": os.path.relpath(path, ROOT), "why": d.get("why")}
        except (OSError, KeyError, ValueError):
            pass
    return None


def live_traffic(kernel_substr, batch, timeout_s=150, extra_env=None):
    """HBM bytes per launch of the dominant kernel, measured IN THIS RUN: two short child runs of this very script under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, no tracing, as MI355X_MICROARCH.md prescribes:
    counter values are KiB per dispatch, FETCH_SIZE reports half of a wide streaming read on gfx950 and is doubled).
    None when rocprofv3 is unavailable or a pass fails -- never a stale constant."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp")
    if extra_env:
        env.update(extra_env)
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        td = tempfile.mkdtemp(prefix="hss_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", ctr, "-d", td, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
                   "--steps", "3", "--warmup", "1", "--settle-steps", "0", "--batch", str(batch)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            dbs = glob.glob(os.path.join(td, "**", "*_results.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None
            db = sqlite3.connect(dbs[0])
            rows = list(db.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? "
                                   "group by kernel_name", (ctr,)))
            hit = [v for name, v, cnt in rows if kernel_substr in name]
            if not hit:
                return None
            vals[ctr] = float(hit[0])
        except Exception:
            return None
        finally:
            shutil.rmtree(td, ignore_errors=True)
    return {"hbm_bytes_per_launch": int(round((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)),
            "FETCH_SIZE_KiB": round(vals["FETCH_SIZE"], 1), "WRITE_SIZE_KiB": round(vals["WRITE_SIZE"], 1),
            "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two child runs of this script (3 steps each) inside this run; "
                   "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 per dispatch"}


def power_leg(run_steps, seconds=2.0):
    """Shader clock and socket power while the workload runs back to back (rocm-smi, sampled from a thread; OUTSIDE the timed region): on
    real data this kernel sits at the chip's power limit and the clock gives way (profiles/r06_team_waits.txt section 4) -- the number that
    explains why the same sources give 165-175 us on different boxes.  None where rocm-smi is missing."""
    import re
    import shutil
    import subprocess
    import threading
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return None
    samples, stop = [], [False]

    def watch():
        while not stop[0]:
            try:
                r = subprocess.run([exe, "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
                sclk = re.search(r"sclk clock level.*\((\d+)Mhz\)", r)
                pw = re.search(r"Power \(W\):\s*([\d.]+)", r)
                if sclk and pw:
                    samples.append((int(sclk.group(1)), float(pw.group(1))))
            except Exception:
                pass
            time.sleep(0.05)
    th = threading.Thread(target=watch, daemon=True)
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        run_steps(200)
    stop[0] = True
    th.join(timeout=10)
    s = samples[1:] if len(samples) > 3 else samples            # (the first sample may still see the idle clock)
    if not s:
        return None
    clk = sorted(v[0] for v in s)
    pw = sorted(v[1] for v in s)
    return {"sclk_mhz": clk[len(clk) // 2], "power_w": round(pw[len(pw) // 2], 1), "power_w_max": round(pw[-1], 1), "samples": len(s),
            "how": f"rocm-smi --showclocks --showpower every ~50 ms over {seconds:.0f} s of the same steps queued back to back, after the timed region; medians"}


def _cpu_leg(X, w, threads, budget_s):
    """Windows per second of the oracle on `threads` OpenMP threads over about `budget_s` seconds of work."""
    import oracle
    oracle.features(X[:threads], 1000, w, (25, 200), "stack", nthreads=threads)      # warm-up (page faults of the scratch)
    done, t0 = 0, time.perf_counter()
    chunk = max(threads * 16, 32) if threads > 1 else 8
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
    return done, el


def _cpu_quota():
    """CPUs this process may really use: the scheduler affinity, cut down by the cgroup's CPU quota where there is one (a container that
    shows 256 host cores and grants 16 of them runs 128 OpenMP threads at the speed of 16: round 5's "2 041 windows/s on 128 threads")."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    return n, quota


def cpu_baseline(X, w, budget_s=12.0):
    """The oracle ("port" of the reference CPU path: fp64 fsst + wrapper epilogue) on the host cores of this box, on a
    bounded sample of the same workload -- SURVEY section 8(d)'s two legs: (ii) OpenMP over windows on the host cores this process is
    granted (`cpu_baseline`: the thread count is the best of a short probe around the cgroup quota, and it is what `cores` says) and
    (i) one thread (`cpu_baseline_1t`, returned as the second object)."""
    import oracle
    oracle.build()
    visible, quota = _cpu_quota()
    top = max(1, min(visible, oracle.max_threads()))
    granted = top if quota is None else max(1, min(top, int(quota + 0.5)))
    cands = sorted({max(1, granted // 2), granted, min(top, 2 * granted), min(top, 4 * granted)})
    probe = {}
    for th in cands:                                       # ~1 s each: which thread count the box really rewards
        d, e = _cpu_leg(X, w, th, 1.0)
        probe[th] = d / e
    threads = max(probe, key=probe.get)
    done, el = _cpu_leg(X, w, threads, budget_s * 0.5)
    d1, e1 = _cpu_leg(X, w, 1, budget_s * 0.25)
    what = "fp64 C restatement (oracle/fsst_oracle.c: two full complex FFTs per frame)"
    allc = {"value": round(done / el, 2), "unit": "windows/s", "cores": threads, "kind": "port",
            "sample": f"{done} of the workload's 2000-sample windows in {el:.1f} s, {what}, OpenMP over windows on {threads} threads "
                      f"({visible} CPUs visible, cgroup quota {'none' if quota is None else f'{quota:.1f}'}; probe windows/s by threads: "
                      f"{ {k: round(v) for k, v in probe.items()} }), per-thread scratch allocated once",
            "per_core": round(done / el / threads, 2)}
    one = {"value": round(d1 / e1, 2), "unit": "windows/s", "cores": 1, "kind": "port",
           "sample": f"{d1} of the workload's 2000-sample windows in {e1:.1f} s, {what}, one thread"}
    return allc, one


def self_launch(ngpus: int) -> int:
    """Re-exec this command under torch.distributed.run with `ngpus` ranks on this node; fails loudly when fewer
    GPUs are visible (never a silent 1-rank run that reports n_gpus = N)."""
    import socket
    import subprocess

    import torch
    have = torch.cuda.device_count()
    if "--dry-run-gloo" in sys.argv:
        have = ngpus if have >= 1 else 0                   # every rank shares GPU 0; the exchange goes over gloo
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


def bench_c5(steps, warmup):
    """BASELINE config 5: 64 channels x 4 kHz, 128 new samples per channel and step, rolling FSST (nwin 512 = the
    same 128 ms window and 7.8125 Hz grid as the canonical configuration), running-moments z-score.  Reports
    device-resident steps/s and the HOST-VISIBLE latency of a step: last sample of a chunk in a host buffer ->
    its (64, 128, 44) features in a host buffer (pinned staging, ONE launch that reads the chunk and writes the features in
    pinned host memory itself, one synchronisation)."""
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
    steps, warm = steps, max(warmup, 20)
    xh = synth.pcg_windows(ch, chunk * 64, fs=fs, seed=2)
    xd = torch.from_numpy(xh).to(dev)
    for i in range(warm):
        st.step(xd[:, (i % 64) * chunk:(i % 64 + 1) * chunk], copy=False)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        st.step(xd[:, (i % 64) * chunk:(i % 64 + 1) * chunk], copy=False)
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
                                        "path": "pinned host chunk (32 KiB), read by the kernel -> one launch -> features (1.44 MB) stored to pinned host memory by the "
                                                "kernel -> stream sync", "kernel": st.last_kernel()},
            "roofline": {"bound": "hbm", "achieved": round((ch * chunk * 4 + ch * chunk * 44 * 4) / dt / 1e9, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round((ch * chunk * 4 + ch * chunk * 44 * 4) / dt / 1e9 / HBM_PEAK_GBS, 6), "traffic": None,
                         "note": "latency bound: 1.47 MB per step, one 16-frame group per pair of waves"}}
    return line


def bench_c1(tf, calls=400):
    """BASELINE config 1 through the UNCHANGED dataset loop: the transform called once per 2000-sample frame with a CPU tensor and
    returning a CPU tensor (/root/reference/hss/datasets/heart_sounds.py:166-168,199-201) -- one window per call, host to host,
    PCIe and Python included -- and the 33 frames of one 35 500-sample recording that way (hss/utils/preprocess.py:40-52)."""
    import numpy as np
    import torch

    from heart_sounds_segmentation_amd import synth
    from heart_sounds_segmentation_amd.framing import frame_batch
    fr = torch.from_numpy(synth.pcg_windows(1, 2000, seed=77)[0]).reshape(2000, 1)
    for _ in range(50):
        tf(fr)
    lat = []
    for _ in range(calls):
        t1 = time.perf_counter()
        y = tf(fr)
        lat.append(time.perf_counter() - t1)
    lat = np.sort(np.asarray(lat)) * 1e3
    rec = torch.from_numpy(synth.recording(35500, seed=78))
    frames = frame_batch(rec, 1000, 2000)                    # the reference's 33 frames
    for _ in range(3):
        for f in frames:
            tf(f.reshape(2000, 1))
    t1 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        feats = [tf(f.reshape(2000, 1)) for f in frames]
    rec_ms = (time.perf_counter() - t1) / reps * 1e3
    # the in-memory dataset (main.py:166, heart_sounds.py:155-169) KEEPS every result: after the pool's 64 lent buffers every call fills a fresh
    # tensor by a copy from pinned memory the GPU has just written (cache-cold for the CPU)
    kept = [tf(fr) for _ in range(80)]
    lat_k = []
    for _ in range(300):
        t1 = time.perf_counter()
        kept.append(tf(fr))
        lat_k.append(time.perf_counter() - t1)
    del kept
    lat_k = np.sort(np.asarray(lat_k)) * 1e3
    return {"metric": "drop-in FSST.__call__, one 2000-sample CPU frame per call (the reference's dataset loop)",
            "value": round(1e3 / float(np.median(lat)), 1), "unit": "windows/s per process",
            "ms_per_call": {"median": round(float(np.median(lat)), 4), "min": round(float(lat[0]), 4), "p99": round(float(lat[int(0.99 * (len(lat) - 1))]), 4)},
            "recording_35500_samples": {"frames": int(frames.shape[0]), "ms": round(rec_ms, 3),
                                        "out_shape": [int(v) for v in feats[0].shape]},
            "ms_per_call_results_kept": {"median": round(float(np.median(lat_k)), 4), "p99": round(float(lat_k[int(0.99 * (len(lat_k) - 1))]), 4),
                                         "note": "every result kept alive (the in-memory dataset): past the 64 lent buffers a call copies its 352 kB out of pinned memory"},
            "path": "CPU float32 (2000, 1) tensor -> pinned mapped staging read by the kernel -> team kernel (one team) -> features stored by the kernel into a "
                    "pinned buffer LENT to the caller as the returned tensor -> the host waits for the word the kernel's last block stores to pinned "
                    "memory (PCIe, Python and the wait included; no stream synchronisation, no copy)",
            "kernel": tf.last_kernel()}


def bench_c3(dev, rank, world, use_dist, steps, warmup, nrec=792, T=35500, host_fed=True):
    """BASELINE config 3: the corpus preprocessing of /root/reference/hss/datasets/heart_sounds.py:155-169 -- every
    recording framed (stride 1000, length 2000: 33 frames per 35 500-sample recording) and every frame transformed --
    on a 792-recording stand-in (26 136 windows; the Springer corpus itself needs a download), RECORDINGS split in
    contiguous blocks over the ranks (framing stays local), frame-list launches of <= 4096 windows
    (hssfsst_exec_list), then ONE ragged all-gather of the feature blocks on the process group's device.  Timed with
    the recordings resident in HBM; strong scaling (the corpus is fixed).  `host_fed`: corpus.build_features from host
    recordings (pinned double-buffered uploads), device-kept and host-returned, on a quarter of the corpus."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from heart_sounds_segmentation_amd import FSST, corpus, dist as hdist, synth
    from heart_sounds_segmentation_amd.framing import frame_starts
    n, stride = 2000, 1000
    w = synth.kaiser_window(128, 0.5)
    tf = FSST(1000, w, truncate_freq=(25, 200), stack=True, device=dev)
    lo, hi = hdist.shard_bounds(nrec, world, rank)
    # stand-in recordings: 8 distinct synthetic recordings, rolled by a recording-specific offset (generating 792
    # independent ones costs a minute of numpy; the transform's cost does not depend on the content)
    base = [synth.recording(T, seed=synth.SEED + 10 + i) for i in range(8)]
    recs = [np.roll(base[i % 8], 97 * i) for i in range(lo, hi)]
    st1 = frame_starts(T, stride, n)[0]
    per = int(st1.shape[0])
    mine = per * len(recs)
    groups, g = [], 0
    gsz = max(1, 4096 // per)                                  # recordings per launch
    xd = torch.from_numpy(np.concatenate(recs) if recs else np.zeros(0, np.float32)).to(dev)
    while g < len(recs):
        k = min(gsz, len(recs) - g)
        starts = torch.from_numpy(np.concatenate([st1 + (g + j) * T for j in range(k)])).to(dev)
        groups.append((g * per, k * per, starts))
        g += k
    arena = torch.empty((mine, n, 44), dtype=torch.float32, device=dev)

    def step():
        for row, cnt, starts in groups:
            tf.frames(xd, starts, n, out=arena[row:row + cnt])

    def sync_all():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(warmup, 1)):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync_all()
    el = time.perf_counter() - t0
    tf.check()
    gather = None
    total = per * nrec
    if use_dist:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        try:
            items = corpus.FrameItems(arena, None)
            full = corpus.gather_features(items)               # warm-up
            ok = tuple(full.shape) == (total, n, 44) and full.is_cuda
            sync_all()
            g0 = time.perf_counter()
            full = corpus.gather_features(items)
            sync_all()
            gt = torch.tensor([time.perf_counter() - g0], dtype=torch.float64, device=dev)
            dist.all_reduce(gt, op=dist.ReduceOp.MAX)
            gms = float(gt.item()) * 1e3
            gather = {"allgather_ms": round(gms, 3), "bytes_total": total * n * 44 * 4, "shape_ok": bool(ok),
                      "value_with_allgather": round(total / (el / steps + gms * 1e-3), 1)}
            del full
        except Exception as e:
            gather = {"error": f"{type(e).__name__}: {e}"[:200]}
    res = {"metric": "PCG windows/sec FSST, corpus preprocessing (C3)", "value": round(total * steps / el, 1), "unit": "windows/s",
           "n_gpus": world, "steps": steps, "warmup": max(warmup, 1), "ms_per_step": round(el / steps * 1e3, 4),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"C3: {nrec} synthetic recordings x {T} samples -> {total} windows of 2000 (stride 1000), "
                                  "recording-level split over the ranks, frame-list launches of <= 4096 windows, Kaiser(128,0.5), "
                                  "band [25,200] Hz, stack=True", "windows_this_rank": mine, "launches_per_step": len(groups),
                      "parallelism": f"recording-sharded x{world}, ragged all-gather of the feature blocks"},
           "roofline": {"bound": "hbm", "achieved": round(total * steps / el * BYTES_PER_WINDOW / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(total * steps / el * BYTES_PER_WINDOW / 1e9 / HBM_PEAK_GBS / max(world, 1), 5), "traffic": None,
                        "note": "whole step (all launches and gaps) over the algorithmic bytes; per GPU"}}
    if gather:
        res["allgather"] = gather
    if host_fed and world == 1:
        q = nrec // 4
        hrecs = [(torch.from_numpy(np.roll(base[i % 8], 97 * i)), None) for i in range(q)]
        out = {}
        builder = corpus.CorpusBuilder(tf, device=dev)
        for keep in (True, False):
            first = builder.build(hrecs, keep_on_device=keep)                 # allocates the staging buffers and the arena
            torch.cuda.synchronize(dev)
            dts = []
            for _ in range(3):                                                # (the first timed call still faults pinned pages in)
                t1 = time.perf_counter()
                items = builder.build(hrecs, keep_on_device=keep, out=first.features)
                torch.cuda.synchronize(dev)
                dts.append(time.perf_counter() - t1)
            dt = min(dts)
            out["device_kept" if keep else "host_returned"] = {"windows_per_s": round(len(items) / dt, 1), "seconds": round(dt, 4),
                                                                "windows": len(items), "calls_timed": len(dts),
                                                                "seconds_each": [round(v, 4) for v in dts]}
            del items, first
        out["note"] = (f"corpus.CorpusBuilder.build from {q} HOST recordings (pageable float32), best of three calls of a builder after the allocating one (staging "
                       "buffers and the feature arena reused): pinned double-buffered uploads on a side stream, frame-list launches "
                       "into the arena; host-returned = group-wise D2H into a pinned host arena on a third stream")
        res["host_fed"] = out
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--settle-steps", type=int, default=300,
                    help="extra untimed steps run BEFORE the W warm-up steps so the GPU reaches its sustained clocks "
                         "(a 20 ms burst from idle measures the clock ramp); counted in the reported `warmup`")
    ap.add_argument("--batch", type=int, default=1024, help="windows per GPU")
    ap.add_argument("--config", choices=("c2", "c3", "c5"), default="c2")
    ap.add_argument("--no-extras", action="store_true", help="c2 only: skip the C3 / C5 sub-measurements and the live PMC passes")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)     # a short c2 run under rocprofv3 --pmc
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--c3-recordings", type=int, default=792, help="recordings of the C3 corpus stand-in (792 = the Springer corpus' count)")
    ap.add_argument("--dry-run-gloo", action="store_true",
                    help="drive the WHOLE N-rank path (self-launch, shard, C2 + C3, gathers, JSON with n_gpus = N) with every rank on GPU 0 and "
                         "the exchange over gloo: the only thing left untested for an N-GPU box is RCCL itself.  Not a measurement.")
    ap.add_argument("--cpu-budget", type=float, default=16.0)
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.config == "c5":
        if args.gpus != 1:
            ap.error("--config c5 is a single-GPU measurement")
        print(json.dumps(bench_c5(args.steps, args.warmup)), flush=True)
        return

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
    if not args.dry_run_gloo and torch.cuda.device_count() < min(world, int(os.environ.get("LOCAL_WORLD_SIZE", world))):
        raise SystemExit(f"bench.py: --gpus {args.gpus} needs {world} visible GPUs, found {torch.cuda.device_count()}")
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if args.dry_run_gloo else int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("HSS_BENCH_FORCE_DIST") == "1"   # the latter: 1-rank RCCL smoke test
    torch.cuda.set_device(local)
    if use_dist:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dry_run_gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local),
                                    timeout=datetime.timedelta(seconds=300))
        world = dist.get_world_size()                      # what RCCL actually sees
    dev = torch.device("cuda", local)

    if args.config == "c3":
        res = bench_c3(dev, rank, world, use_dist, max(1, min(args.steps, 20)), min(args.warmup, 3), nrec=args.c3_recordings)
        if args.dry_run_gloo:
            res["config"]["dry_run"] = "every rank on GPU 0, exchange over gloo: a path test, not a measurement"
        if rank == 0:
            print(json.dumps(res), flush=True)
        if use_dist:
            dist.destroy_process_group()
        return

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
    # Events are packets on the exec stream: the library's pair around the kernel of EVERY step plus a marker per step cost ~8 us of a
    # 0.19 ms step (measured: profiles/r05_team_diet.txt).  So the dominant kernel is timed on 5-10 of the K steps (its duration does not
    # depend on which), and the spread of the steps comes from as many markers; the timed region itself is still the K steps between the barriers.
    every = max(1, args.steps // (10 if args.steps >= 100 else 5))
    tf.set_timing(every, local)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps // every + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        tf.batch(X, out=out)
        if (i + 1) % every == 0:
            marks[(i + 1) // every].record()
    sync_all()
    elapsed = time.perf_counter() - t0
    seg_ms = sorted(marks[i].elapsed_time(marks[i + 1]) / every for i in range(args.steps // every)) if args.steps >= every else []
    core_ms, norm_ms, ncalls = tf.timing(local)
    tf.set_timing(False, local)
    fused = tf.check(local)                                   # raises if a kernel reported a failed internal wait
    if args.pmc_child:                                        # (under rocprofv3 --pmc: the parent reads the counters)
        return
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
        # which kernel ran: asked of the library (hssfsst_plan_last_kernel), not assumed
        klong = tf.last_kernel(local)
        kname = klong.split(" [")[0].split(" teams of")[0]
        kdesc = {1: " (transform + z-score in one launch: one CU per signal, tile round-trips through HBM inside the launch)",
                 2: " (transform + z-score in one launch: teams of CUs, four waves per SIMD, features z-scored in registers and written once)"}.get(
                     fused, " (transform; z-score is a second kernel)")
        prof = pmc_profile(kname) if B == 1024 else None
        live = live_traffic(kname, B) if (world == 1 and not args.no_extras) else None
        traffic = live["hbm_bytes_per_launch"] if live else (prof.get("hbm_bytes_per_launch") if prof else None)
        roof = {"bound": "hbm", "kernel": kname + kdesc, "launch": klong,
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "traffic_source": (live["how"] if live else (prof["_file"] + " (same sources, SHA-256 checked)" if prof else None)),
                "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(dom_ms, 4),
                # everything of a step that is not the dominant kernel: the library's own events (z-score kernels of the two-launch
                # path) and -- wall clock minus kernel -- the gated fallback launch queued behind every team launch (~4 us) and gaps
                "other_kernels_avg_ms": round(max(norm_ms / max(ncalls, 1), ms_per_step - dom_ms, 0.0), 4), "launches_timed": ncalls,
                "launches_timed_of": args.steps,
                "step_ms_spread": ({"min": round(seg_ms[0], 4), "median": round(seg_ms[len(seg_ms) // 2], 4), "max": round(seg_ms[-1], 4),
                                    "samples": len(seg_ms), "steps_per_sample": every,
                                    "how": "HIP events on the exec stream inside the timed region, one per steps_per_sample steps"} if seg_ms else None),
                # the whole path (every kernel of a step + gaps), the figure north_star's 40 % is about
                "step_achieved": round(step_gbs, 2), "step_frac": round(step_gbs / HBM_PEAK_GBS, 5)}
        if B == 1024 and "team16" in kname:
            roof["box_spread"] = box_spread()
        if world == 1 and not args.no_extras:
            def _steps(k):
                for _ in range(k):
                    tf.batch(X, out=out)
                torch.cuda.synchronize()
            pw = power_leg(_steps)
            if pw:
                roof["sclk_mhz"], roof["power_w"] = pw["sclk_mhz"], pw["power_w"]
                roof["power"] = pw
                roof["power_note"] = ("on real data the kernel runs at the chip's power limit (~1 400 W) and the shader clock gives way (~2.2 GHz of 2.4): "
                                      "a window's ENERGY bounds this path before its bytes do (profiles/r06_team_waits.txt)")
        if prof and all(k in prof for k in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES")) and prof["SQ_WAVE_CYCLES"] > 0:
            # the bound that actually binds: issue slots.  A wave64 VALU instruction holds its SIMD for 4 clocks, a
            # v_mfma_f32_16x16x32_f16 for 16 (they do not overlap: profiles/r01_mfma_valu_overlap_ubench.txt); SQ_WAVE_CYCLES counts 4-clock
            # units per wave and four waves share a SIMD for the whole launch, so it is also the launch's SIMD-clocks summed over SIMDs
            roof["issue_frac"] = round((4.0 * prof["SQ_INSTS_VALU"] + 16.0 * prof["SQ_INSTS_MFMA"]) / prof["SQ_WAVE_CYCLES"], 4)
            roof["issue_note"] = ("(4 x VALU + 16 x MFMA wave-instructions) / SIMD-clocks of the launch, from " + prof["_file"] +
                                  ": the kernel is issue-bound, this is the fraction that can reach 1.0; `frac` (HBM) stays the headline")
        if prof and "SQ_INSTS_MFMA" in prof and "SQ_INSTS_VALU" in prof and dom_ms > 0:
            canon = "canon" in kname or "4, 22" in kname
            flop = (0 if canon else prof["SQ_INSTS_MFMA"] * MFMA_FLOP) + prof["SQ_INSTS_VALU"] * 64 * VALU_FLOP_PER_LANE
            roof["fp32_tflops"] = round(flop / (dom_ms * 1e-3) / 1e12, 2)
            roof["fp32_frac"] = round(flop / (dom_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)
            if canon:
                roof["mfma_f16_tflops"] = round(prof["SQ_INSTS_MFMA"] * MFMA_F16_FLOP / (dom_ms * 1e-3) / 1e12, 2)
                roof["fp32_note"] = (f"{int(prof['SQ_INSTS_VALU'])} VALU wave-instructions x 64 lanes x {VALU_FLOP_PER_LANE} FLOP (estimate) per launch, "
                                     f"peak {FP32_PEAK_TFLOPS} TFLOP/s; besides {int(prof['SQ_INSTS_MFMA'])} v_mfma_f32_16x16x32_f16 x {MFMA_F16_FLOP} FLOP "
                                     "(the window fold with split operands: 4 half products per real one) on the 16-bit matrix pipe (" + prof["_file"] + ")")
            else:
                roof["fp32_note"] = (f"{int(prof['SQ_INSTS_MFMA'])} v_mfma_f32_16x16x4_f32 x {MFMA_FLOP} FLOP + "
                                     f"{int(prof['SQ_INSTS_VALU'])} VALU wave-instructions x 64 lanes x {VALU_FLOP_PER_LANE} FLOP (estimate) "
                                     f"per launch ({prof['_file']}); peak {FP32_PEAK_TFLOPS} TFLOP/s")
        # the same workload on the other single-launch kernel (one CU per signal: the tile round-trips through HBM), beside the default
        if world == 1 and not args.no_extras and fused == 2 and not os.environ.get("HSSFSST_NO_CANON"):
            try:
                tf.set_zpath("one_cu", local)
                for _ in range(50):
                    tf.batch(X, out=out)
                torch.cuda.synchronize(dev)
                tf.set_timing(True, local)
                for _ in range(min(args.steps, 500)):
                    tf.batch(X, out=out)
                t_ms, _, t_n = tf.timing(local)
                tf.set_timing(False, local)
                t_path = tf.check(local)
                o_name = tf.last_kernel(local).split(" [")[0]
                tf.set_zpath("auto", local)
                if t_path == 1 and t_n > 0:
                    tl = live_traffic(o_name, B, extra_env={"HSSFSST_NO_TEAM": "1"})
                    t_avg = t_ms / t_n
                    roof["one_cu_kernel"] = {"kernel": o_name + " (transform + z-score in one launch: one CU per signal, the un-normalised tile makes a "
                                                                "round trip through HBM; bit-identical output)",
                                             "avg_launch_ms": round(t_avg, 4), "achieved": round(alg / (t_avg * 1e-3) / 1e9, 2),
                                             "frac": round(alg / (t_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                             "traffic": tl["hbm_bytes_per_launch"] if tl else None,
                                             "traffic_over_algorithmic": round(tl["hbm_bytes_per_launch"] / alg, 3) if tl else None}
            except Exception as e:                               # never lose the bench line to the side measurement
                roof["one_cu_kernel"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        if traffic:
            roof["traffic_over_algorithmic"] = round(traffic / alg, 3)
        line = {
            "metric": "PCG windows/sec FSST (1 kHz, 2000-sample)", "value": round(value, 1),
            "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C2: {B} x {n} fp32 synthetic PCG windows per GPU, fs=1000, "
                                   "Kaiser(128,0.5), band [25,200] Hz, stack=True -> (2000,44) fp32",
                       "windows_per_gpu": B, "clock_settle_steps": max(args.settle_steps, 0),
                       "untimed_steps_total": untimed,
                       "parallelism": f"window-sharded x{world}, no data-path collective",
                       "arithmetic": "float32 throughout; the window fold runs on the f16 matrix pipe with split operands "
                                     "(sample and constant each a pair of halves = 22 bits, fp32 accumulation), every rounding "
                                     "decision float32 cannot make in float64",
                       "arithmetic_max_rel_err": census_max_rel_err(),
                       "zscore": {1: "same launch (one CU per signal)", 2: "same launch (team kernel)"}.get(fused, "second kernel")},
            "roofline": roof,
        }
        if gather:
            line["allgather"] = gather
        if args.dry_run_gloo:
            line["config"]["dry_run"] = "every rank on GPU 0, exchange over gloo: a path test, not a measurement"
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], line["cpu_baseline_1t"] = cpu_baseline(Xh, w, args.cpu_budget)
    # the other BASELINE configurations of the path, carried by the same line: C3 (corpus preprocessing, recording-level
    # split, RCCL all-gather when N > 1) on every N, C5 (streaming) on one GPU
    extras = {}
    if not args.no_extras and world == 1:
        try:
            extras["c1"] = bench_c1(tf)
        except Exception as e:                                   # never lose the bench line to a side measurement
            extras["c1"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if not args.no_extras:
        del out
        try:
            c3 = bench_c3(dev, rank, world, use_dist, 3, 1, nrec=args.c3_recordings)
            extras["c3"] = {k: c3[k] for k in ("metric", "value", "unit", "ms_per_step", "scaling", "config", "allgather", "host_fed", "roofline") if k in c3}
        except Exception as e:                                   # never lose the bench line to a side measurement
            extras["c3"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1:
            try:
                c5 = bench_c5(2000, 100)
                extras["c5"] = {k: c5[k] for k in ("metric", "value", "unit", "ms_per_step", "config", "latency_host_visible_ms")}
            except Exception as e:
                extras["c5"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0:
        line.update(extras)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
