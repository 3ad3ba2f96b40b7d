#!/usr/bin/env python3
"""N PROCESSES sharing one GPU, each running the reference's dataset loop -- one 2000-sample CPU frame per FSST.__call__
(/root/reference/hss/datasets/heart_sounds.py:166-168; /root/reference/main.py:206 forks os.cpu_count() DataLoader workers) -- for a fixed time:
windows/s per process and in all, the share of calls whose team launch gave itself up, median / p99 of a call, every 16th result compared with
the two-launch path's.  usage: share_curve.py [procs ...]   (default 1 3 8 32)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, seconds, q, go):
    import torch
    torch.set_num_threads(1)                             # (as torch's DataLoader workers do)
    from heart_sounds_segmentation_amd import FSST, synth
    w = synth.kaiser_window(128, 0.5)
    tf = FSST(1000, w, truncate_freq=(25, 200), stack=True)
    ref = FSST(1000, w, truncate_freq=(25, 200), stack=True)
    ref.set_zpath("two_launch")
    X = torch.from_numpy(synth.pcg_windows(64, 2000, seed=500 + rank))
    frames = [X[i].reshape(2000, 1).contiguous() for i in range(64)]
    want = [None] * 64
    for i in (0, 21, 42, 63):                            # (the checked frames' two-launch results, made before the clock starts)
        want[i] = ref.batch(X[i:i + 1].cuda())[0].cpu()
    tf(frames[0]); torch.cuda.synchronize()
    go.wait()
    lat, bad, calls, t0 = [], 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        i = calls & 63
        a = time.perf_counter()
        got = tf(frames[i])
        lat.append(time.perf_counter() - a)
        if want[i] is not None:                          # (every 16th call)
            bad += not torch.equal(got, want[i])
        calls += 1
    el = time.perf_counter() - t0
    lat = np.asarray(lat) * 1e3
    q.put((rank, calls, el, tf.fallbacks(), bad, float(np.median(lat)), float(np.percentile(lat, 99)), float(lat.max())))


if __name__ == "__main__":
    import torch.multiprocessing as mp
    counts = [int(a) for a in sys.argv[1:]] or [1, 3, 8, 32]
    seconds = float(os.environ.get("SHARE_SECONDS", "4"))
    ctx = mp.get_context("spawn")
    rc = 0
    for procs in counts:
        q = ctx.Queue()
        go = ctx.Barrier(procs)
        ps = [ctx.Process(target=worker, args=(r, seconds, q, go)) for r in range(procs)]
        for p in ps: p.start()
        res, t_end = [], time.time() + 180 + 4 * procs
        while len(res) < len(ps) and time.time() < t_end:
            try:
                res.append(q.get(timeout=1.0))
            except Exception:
                if any(p.exitcode not in (None, 0) for p in ps):
                    break
        for p in ps:
            if p.is_alive() and len(res) < len(ps): p.terminate()
            p.join(10)
        if len(res) < procs:
            print(f"{procs:3d} processes: only {len(res)} finished; exit codes {[p.exitcode for p in ps]}")
            rc = 1
            continue
        calls = sum(r[1] for r in res); rate = sum(r[1] / r[2] for r in res)
        fb = sum(r[3] for r in res); bad = sum(r[4] for r in res)
        print(f"{procs:3d} processes: {rate:9.0f} windows/s in all, {rate / procs:8.0f} per process (min {min(r[1] / r[2] for r in res):.0f}, max {max(r[1] / r[2] for r in res):.0f}); "
              f"team launches given up {fb} of {calls} calls ({100.0 * fb / max(calls, 1):.2f} %); call median {np.median([r[5] for r in res]):.3f} ms, "
              f"p99 {np.max([r[6] for r in res]):.3f} ms, max {np.max([r[7] for r in res]):.1f} ms; {bad} results differ from the two-launch path", flush=True)
        rc |= bad != 0
    sys.exit(rc)
