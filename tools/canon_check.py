#!/usr/bin/env python3
"""Development check of one build of libhssfsst.so on the canonical configuration (Kaiser(128, 0.5), [25, 200] Hz, STACK):
parity against the oracle on several inputs and shapes (reports max rel err), bit-identity of the z-score paths
(fused / team / two-launch via HSSFSST_NO_FUSED in a child process), and time per 1024-window exec.
usage: canon_check.py lib.so [quick]"""
import ctypes
import os
import subprocess
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heart_sounds_segmentation_amd import synth  # noqa: E402


def load(path):
    L = ctypes.CDLL(path)
    vp, ip, dp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)
    L.hssfsst_plan_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.c_int, dp, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int]
    L.hssfsst_exec.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp]
    L.hssfsst_plan_set_timing.argtypes = [vp, ctypes.c_int]
    L.hssfsst_plan_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ip]
    L.hssfsst_plan_check.argtypes = [vp]
    L.hssfsst_plan_last_exec_fused.argtypes = [vp]
    L.hssfsst_last_error.restype = ctypes.c_char_p
    return L


def run(L, plan, X, K2=44):
    B, n = X.shape
    Xd = torch.from_numpy(X).cuda()
    out = torch.full((B, n, K2), float("nan"), dtype=torch.float32, device="cuda")
    rc = L.hssfsst_exec(plan, ctypes.c_void_p(Xd.data_ptr()), B, n, 1, ctypes.c_void_p(out.data_ptr()), 1, None)
    assert rc == 0, L.hssfsst_last_error()
    rc = L.hssfsst_plan_check(plan)
    assert rc == 0, L.hssfsst_last_error()
    return out.cpu().numpy(), L.hssfsst_plan_last_exec_fused(plan)


def inputs(kind, B, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "pcg":
        return synth.pcg_windows(B, n, seed=seed)
    if kind == "noise":
        return rng.standard_normal((B, n)).astype(np.float32)
    if kind == "pcg_x8":
        return (synth.pcg_windows(B, n, seed=seed) * 8.0).astype(np.float32)
    if kind == "pcg_tiny":
        return (synth.pcg_windows(B, n, seed=seed) * 1e-6).astype(np.float32)
    if kind == "burst":                                   # silence with one loud burst: the tile scale varies 1e6 along the signal
        x = (1e-6 * rng.standard_normal((B, n))).astype(np.float32)
        x[:, n // 2: n // 2 + 150] += (1000.0 * np.sin(2 * np.pi * 60 * np.arange(150) / 1000.0)).astype(np.float32)
        return x
    if kind == "dc":
        return (synth.pcg_windows(B, n, seed=seed) + 3.0).astype(np.float32)
    if kind == "tone":
        t = np.arange(n) / 1000.0
        return np.tile(np.cos(2 * np.pi * 101.3 * t).astype(np.float32), (B, 1))
    raise ValueError(kind)


def main():
    path = sys.argv[1]
    quick = len(sys.argv) > 2 and sys.argv[2] == "quick"
    child = os.environ.get("CANON_CHECK_CHILD")
    L = load(path)
    dp = ctypes.POINTER(ctypes.c_double)
    w = np.ascontiguousarray(synth.kaiser_window(128, 0.5))
    wk = os.environ.get("CANON_WINDOW", "kaiser")
    if wk == "hann": w = np.ascontiguousarray(np.hanning(128).astype(np.float64))
    if wk == "blackman": w = np.ascontiguousarray(np.blackman(128).astype(np.float64))
    if wk == "kaiser10": w = np.ascontiguousarray(np.kaiser(128, 10.0).astype(np.float64))
    plan = ctypes.c_void_p()
    rc = L.hssfsst_plan_create(ctypes.byref(plan), 0, 128, w.ctypes.data_as(dp), 1000.0, 1, 25.0, 200.0, 2)
    assert rc == 0, L.hssfsst_last_error()
    shapes = [(4, 2000), (300, 2000), (1024, 2000), (7, 1999), (33, 1000), (2, 4000), (1, 35500)]
    if child:                                             # print the CRC of every shape's output and leave
        for (B, n) in shapes:
            o, fz = run(L, plan, inputs("pcg", B, n, B + n))
            print(f"crc {B}x{n} {zlib.crc32(o.tobytes())} zpath={fz}", flush=True)
        return
    import oracle
    from tests import parity
    worst = 0.0
    for kind in ["pcg", "noise", "pcg_x8", "pcg_tiny", "burst", "dc", "tone"]:
        for (B, n) in ([(6, 2000)] if quick else [(6, 2000), (3, 1999), (2, 4000)]):
            X = inputs(kind, B, n, 17 + n)
            got, fz = run(L, plan, X)
            ref, hd = oracle.features(X, 1000, w, (25, 200), "stack", nthreads=8, return_halfdist=True)
            rels = []
            for b in range(B):
                try:
                    r = parity.check(got[b], ref[b], hd[b], 0, what=f"{kind}[{b}]")
                    rels.append(r["rel"])
                except AssertionError as e:
                    print("  FAIL", str(e)[:200], flush=True)
                    rels.append(float("nan"))
            worst = max(worst, np.nanmax(rels))
            print(f"parity {kind:9s} {B}x{n}: max rel err {np.nanmax(rels):.2e}  (zpath={fz}, fails={int(np.isnan(rels).sum())})", flush=True)
    print(f"worst rel err vs oracle {worst:.2e} (gate 1e-4)")
    # bit-identity across z-score paths
    mine = {}
    for (B, n) in shapes:
        o, fz = run(L, plan, inputs("pcg", B, n, B + n))
        mine[(B, n)] = (zlib.crc32(o.tobytes()), fz)
    for env in [{"HSSFSST_NO_FUSED": "1"}, {"HSSFSST_NO_TEAM": "1"}, {"HSSFSST_TEAM_ONLY": "1"}]:
        e = dict(os.environ); e.update(env); e["CANON_CHECK_CHILD"] = "1"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), path], env=e, capture_output=True, text=True)
        if r.returncode != 0:
            print("child failed", env, r.stderr[-400:]); continue
        bad = 0
        for line in r.stdout.splitlines():
            if not line.startswith("crc"): continue
            _, shp, crc, zp = line.split()
            B, n = (int(v) for v in shp.split("x"))
            same = int(crc) == mine[(B, n)][0]
            bad += not same
            print(f"  {list(env)[0]:18s} {shp:10s} {zp:8s} vs zpath={mine[(B, n)][1]}: {'bit-identical' if same else 'DIFFERENT'}")
        print(f"{list(env)[0]}: {bad} shapes differ")
    # timing
    X = torch.from_numpy(synth.pcg_windows(1024, 2000)).cuda()
    out = torch.empty((1024, 2000, 44), dtype=torch.float32, device="cuda")
    for rd in range(4):
        L.hssfsst_plan_set_timing(plan, 1)
        for _ in range(200):
            L.hssfsst_exec(plan, ctypes.c_void_p(X.data_ptr()), 1024, 2000, 1, ctypes.c_void_p(out.data_ptr()), 1, None)
        ms = (ctypes.c_float * 2)(); cnt = ctypes.c_int()
        L.hssfsst_plan_timing(plan, ms, ctypes.byref(cnt))
        t = (ms[0] + ms[1]) / cnt.value
    print(f"C2 1024x2000: {t:.4f} ms/exec  {1024 / t / 1e3:.3f} Mwin/s  {368.64e6 / (t * 1e-3) / 8e12 * 100:.2f}% of 8 TB/s  zpath={L.hssfsst_plan_last_exec_fused(plan)}")


if __name__ == "__main__":
    main()
