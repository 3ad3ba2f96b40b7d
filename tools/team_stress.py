#!/usr/bin/env python3
"""Several PROCESSES sharing one GPU, each running team-kernel execs of random small batches back to back (the DataLoader-worker
situation, /root/reference/main.py:202-218): no exec may report a failed wait, and every result must equal the two-launch path's.
usage: team_stress.py [procs=4] [iters=300]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, iters, q, go):
    from heart_sounds_segmentation_amd import FSST, synth
    w = synth.kaiser_window(128, 0.5)
    tf = FSST(1000, w, truncate_freq=(25, 200), stack=True)
    ref = FSST(1000, w, truncate_freq=(25, 200), stack=True)
    ref.set_zpath("two_launch")
    rng = np.random.default_rng(rank)
    X = torch.from_numpy(synth.pcg_windows(320, 2000, seed=100 + rank)).cuda()
    tf.batch(X[:8]); ref.batch(X[:8]); torch.cuda.synchronize()      # plans, kernels, clocks: then all start together
    go.wait()
    bad, teams, t0 = 0, 0, time.time()
    for it in range(iters):
        B = int(rng.integers(1, 320))
        got = tf.batch(X[:B])
        if it % 10 == 0:
            path = tf.check()                           # raises on a failed wait
            teams += path == 2
            want = ref.batch(X[:B])
            bad += not torch.equal(got, want)
    tf.check()
    q.put((rank, bad, teams, time.time() - t0, tf.fallbacks()))


if __name__ == "__main__":
    import torch.multiprocessing as mp
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    go = ctx.Barrier(procs)
    ps = [ctx.Process(target=worker, args=(r, iters, q, go)) for r in range(procs)]
    for p in ps: p.start()
    res, t_end = [], time.time() + 240
    while len(res) < len(ps) and time.time() < t_end:
        try:
            res.append(q.get(timeout=1.0))
        except Exception:
            if any(p.exitcode not in (None, 0) for p in ps):       # a worker died (a failed wait raises): stop at once
                break
    for p in ps:
        if p.is_alive() and len(res) < len(ps): p.terminate()
        p.join(10)
    for r in sorted(res): print(f"proc {r[0]}: {r[1]} mismatches, {r[2]} team-path checks, {r[3]:.1f} s, {r[4]} launches fell back to two launches")
    print("exit codes", [p.exitcode for p in ps])
    sys.exit(1 if any(r[1] for r in res) or any(p.exitcode for p in ps) else 0)
