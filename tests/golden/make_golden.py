#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.  Runs ONLY in the build container (needs /root/reference).

The reference's Python files are imported here -- never copied -- to produce input/output vectors:

* ``hss/transforms/synchrosqueeze.py`` (class FSST) is loaded by path with ``sys.modules['ssq']``
  set to a shim around the CPU oracle (the real ``ssq`` 0.1.0 native package is not obtainable,
  SURVEY.md section 8c).  The outputs therefore pin the WRAPPER (casts, band truncation, abs,
  stack/z-score, orientation: synchrosqueeze.py:50-111) applied to the oracle's ``s, f, t``;
  the core itself stays unpinned (see oracle/fsst_oracle.c header).
* ``hss/utils/preprocess.py`` (frame_signal) -> frame start indices and shapes.
* ``hss/moments/__init__.py`` (update_mean / update_variance) -> scalar sequences.
* ``hss/transforms/resample.py`` (class Resample, scipy.signal.resample underneath) -> resampled signals and the
  label rule of ``hss/datasets/heart_sounds.py:205-206``; the input of the reference's own test
  (``test/test_transforms.py:12-14``) is one of the cases.  This pins oracle/resample_numpy.py to the real reference.

The committed fixtures are data only (inputs + expected outputs).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from heart_sounds_segmentation_amd import synth  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    shim = types.ModuleType("ssq")
    shim.fsst = lambda x, fs, window: oracle.fsst(x, fs, window)
    sys.modules["ssq"] = shim
    ref_ss = _load("ref_synchrosqueeze", os.path.join(REF, "hss/transforms/synchrosqueeze.py"))
    ref_pre = _load("ref_preprocess", os.path.join(REF, "hss/utils/preprocess.py"))
    ref_mom = _load("ref_moments", os.path.join(REF, "hss/moments/__init__.py"))

    kaiser = synth.kaiser_window(128, 0.5)
    from scipy.signal import get_window
    assert np.allclose(kaiser, get_window(("kaiser", 0.5), 128, fftbins=False), rtol=0, atol=1e-15)
    hann64 = get_window("hann", 64, fftbins=False)
    rng = np.random.default_rng(7)

    cases = {}

    def add(tag, x, fs, window, **kw):
        xt = torch.from_numpy(np.ascontiguousarray(x))
        tf = ref_ss.FSST(fs, window, **kw)
        y = tf(xt)
        y = y.contiguous().numpy() if not y.is_complex() else y.contiguous().numpy()
        cases[f"{tag}__x"] = np.asarray(x)
        cases[f"{tag}__fs"] = np.float64(fs)
        cases[f"{tag}__window"] = np.asarray(window, dtype=np.float64)
        cases[f"{tag}__y"] = y
        cases[f"{tag}__abs"] = np.bool_(kw.get("abs", False))
        cases[f"{tag}__stack"] = np.bool_(kw.get("stack", False))
        tr = kw.get("truncate_freq")
        cases[f"{tag}__band"] = np.asarray(tr if tr else (np.nan, np.nan), dtype=np.float64)
        print(tag, x.shape, x.dtype, "->", y.shape, y.dtype)

    # A: the canonical configuration of main.py:153-158 on a dataset-shaped (2000, 1) frame
    xa = synth.pcg_windows(1, 2000, seed=11)[0].reshape(2000, 1)
    add("A_canonical_stack", xa, 1000, kaiser, truncate_freq=(25, 200), stack=True)
    # B: abs wins over stack (synchrosqueeze.py:59-63)
    add("B_abs", rng.standard_normal(600).astype(np.float32), 1000, kaiser,
        truncate_freq=(25, 200), abs=True, stack=True)
    # C/D: raw complex, without and with truncation
    xc = rng.standard_normal(400).astype(np.float32)
    add("C_raw_full", xc, 1000, kaiser)
    add("D_raw_band", xc, 1000, kaiser, truncate_freq=(25, 200))
    # E: stack without truncation (all 65 rows)
    add("E_stack_full", rng.standard_normal(256).astype(np.float32), 1000, kaiser, stack=True)
    # F: another window / rate / band, float64 1-D input as scripts/visualize_signals.py:10-14
    add("F_hann64_f64", rng.standard_normal(500), 2000, hann64, truncate_freq=(100, 600), stack=True)
    # G: abs without truncation
    add("G_abs_full", rng.standard_normal(300).astype(np.float32), 1000, kaiser, abs=True)
    np.savez_compressed(os.path.join(HERE, "fsst_wrapper.npz"), **cases)

    # ValueError contract of _truncate_frequencies (synchrosqueeze.py:104-105)
    try:
        ref_ss.FSST(1000, kaiser)._truncate_frequencies(torch.zeros(65, 4, dtype=torch.complex64),
                                                       torch.zeros(65))
        raise SystemExit("expected ValueError")
    except ValueError as e:
        print("ValueError ok:", e)

    # frame_signal (hss/utils/preprocess.py:7-58)
    fr = {}
    for T in (35000, 35500, 4000, 3000, 2500, 2000, 1500):
        x = torch.arange(T, dtype=torch.float32)
        y = torch.arange(T, dtype=torch.int64) % 4
        frames, labels = ref_pre.frame_signal(x, y, 1000, 2000)
        fr[f"T{T}__starts"] = np.asarray([int(f[0, 0]) for f in frames], dtype=np.int64)
        fr[f"T{T}__lens"] = np.asarray([f.shape[0] for f in frames], dtype=np.int64)
        fr[f"T{T}__cols"] = np.asarray([f.shape[1] for f in frames], dtype=np.int64)
        fr[f"T{T}__label0"] = np.asarray([int(l[0, 0]) for l in labels], dtype=np.int64)
        print("frame_signal", T, len(frames), fr[f"T{T}__lens"][:2])
    np.savez_compressed(os.path.join(HERE, "frame_signal.npz"), **fr)

    # moments (hss/moments/__init__.py:1-36): running mean + Welford M2 over a sequence
    xs = rng.standard_normal(64) * 3.0 + 1.5
    m, var = 0.0, 0.0
    means, m2s = [], []
    for k, xv in enumerate(xs, start=1):
        var = ref_mom.update_variance(float(xv), m, var, k)   # uses the OLD mean
        m = ref_mom.update_mean(m, float(xv), k)
        means.append(m)
        m2s.append(var)
    np.savez_compressed(os.path.join(HERE, "moments.npz"), xs=xs, means=np.asarray(means),
                        m2s=np.asarray(m2s))
    print("moments final", means[-1], m2s[-1] / 63, xs.mean(), xs.var(ddof=1))

    # Resample (hss/transforms/resample.py:5-21) and the label rule of heart_sounds.py:205-206
    ref_rs = _load("ref_resample", os.path.join(REF, "hss/transforms/resample.py"))
    rs = {}
    def add_rs(tag, x, num, dtype=torch.float32):
        y = ref_rs.Resample(num)(x, dtype) if dtype is not torch.float32 else ref_rs.Resample(num)(x)
        rs[f"{tag}__x"] = x.numpy(); rs[f"{tag}__num"] = np.int64(num); rs[f"{tag}__y"] = y.numpy()
        print("resample", tag, tuple(x.shape), x.dtype, "->", tuple(y.shape), y.dtype)
    add_rs("ref_test_input_100", torch.tensor([1, 2, 3, 5]), 100)          # test/test_transforms.py:12-14, num = 100
    add_rs("ref_test_3_to_50", torch.tensor([1, 2, 3]), 50)                # test/test_transforms.py:33-42
    add_rs("ref_test_5_to_50", torch.tensor([1, 2, 3, 4, 5]), 50)
    add_rs("down_even_even", torch.from_numpy(rng.standard_normal(64)), 20, torch.float64)
    add_rs("down_even_odd", torch.from_numpy(rng.standard_normal(64)), 21, torch.float64)
    add_rs("down_odd_even", torch.from_numpy(rng.standard_normal(75)), 30, torch.float64)
    add_rs("up_even_odd", torch.from_numpy(rng.standard_normal(40)), 97, torch.float64)
    add_rs("up_odd_even", torch.from_numpy(rng.standard_normal(41)), 128, torch.float64)
    add_rs("same_len", torch.from_numpy(rng.standard_normal(50)), 50, torch.float64)
    add_rs("to_one", torch.from_numpy(rng.standard_normal(9)), 1, torch.float64)
    add_rs("from_one", torch.from_numpy(rng.standard_normal(1)), 7, torch.float64)
    add_rs("pcg_f32_1000_to_500", torch.from_numpy(synth.pcg_windows(1, 1000, seed=3)[0]), 500)
    # label path: labels 1..4 in runs, as in the corpus; y_new = round(Resample(y)) - 1
    lab = torch.from_numpy(np.repeat(np.tile(np.array([1, 2, 3, 4], dtype=np.int64), 9), 71)[:2500])
    t = ref_rs.Resample(1250)
    rs["labels__y"] = lab.numpy(); rs["labels__num"] = np.int64(1250)
    rs["labels__out"] = (torch.round(t(lab)).type(torch.int64) - 1).numpy()
    rs["labels__raw"] = t(lab, torch.float64).numpy()
    np.savez_compressed(os.path.join(HERE, "resample.npz"), **rs)

    # consumer of the features (hss/model/segmenter.py:5-87) at a reduced hidden size: weights, the
    # non-persistent random h0/c0, an input and the reference module's output (eval mode)
    ref_seg = _load("ref_segmenter", os.path.join(REF, "hss/model/segmenter.py"))
    torch.manual_seed(1234)
    m = ref_seg.HeartSoundSegmenter(input_size=44, batch_size=3, hidden_size=12)
    m.eval()
    xin = torch.randn(3, 40, 44)
    with torch.no_grad():
        yout = m(xin)
    seg = {f"sd__{k}": v.numpy() for k, v in m.state_dict().items()}
    seg.update(h0=m.h0.numpy(), c0=m.c0.numpy(), x=xin.numpy(), y=yout.numpy())
    np.savez_compressed(os.path.join(HERE, "segmenter.npz"), **seg)
    print("segmenter", sorted(m.state_dict().keys())[:3], yout.shape)


def segmenter_c4():
    """BASELINE config C4 at its real size: the reference pipeline end to end on the CPU -- the reference FSST class
    (wrapper over the oracle core, main.py:153-158 configuration) on 50 synthetic 2000-sample windows, then the
    reference ``HeartSoundSegmenter(input_size=44, batch_size=50)`` (hidden 240, main.py:170,221) in eval mode.
    The 1.94 M weights are NOT stored (7.7 MB of noise): they are a function of the seed and of the construction
    order of the reference class (h0, c0, lstm_1, lstm_2, linear -- segmenter.py:38-67), which the test replays;
    a SHA-256 of the state_dict pins that the replay produced the same numbers."""
    import hashlib
    shim = types.ModuleType("ssq")
    shim.fsst = lambda x, fs, window: oracle.fsst(x, fs, window)
    sys.modules["ssq"] = shim
    ref_ss = _load("ref_synchrosqueeze", os.path.join(REF, "hss/transforms/synchrosqueeze.py"))
    ref_seg = _load("ref_segmenter", os.path.join(REF, "hss/model/segmenter.py"))
    seed, wseed, B, n = 4242, 44, 50, 2000
    kaiser = synth.kaiser_window(128, 0.5)
    X = synth.pcg_windows(B, n, seed=wseed)
    tf = ref_ss.FSST(1000, kaiser, truncate_freq=(25, 200), stack=True)
    feats = torch.stack([tf(torch.from_numpy(X[b]).reshape(n, 1)).contiguous() for b in range(B)])
    torch.manual_seed(seed)
    m = ref_seg.HeartSoundSegmenter(input_size=44, batch_size=B)
    m.eval()
    with torch.no_grad():
        y = m(feats)
    h = hashlib.sha256()
    for k in sorted(m.state_dict().keys()):
        h.update(k.encode())
        h.update(m.state_dict()[k].contiguous().numpy().tobytes())
    h.update(m.h0.numpy().tobytes())
    h.update(m.c0.numpy().tobytes())
    np.savez_compressed(os.path.join(HERE, "segmenter_c4.npz"), y=y.numpy(), seed=np.int64(seed), window_seed=np.int64(wseed),
                        sha256=np.frombuffer(h.digest(), dtype=np.uint8), feat_probe=feats[:, ::250, :].numpy())
    print("segmenter_c4", tuple(feats.shape), "->", tuple(y.shape), h.hexdigest()[:16])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "c4":
        segmenter_c4()
        sys.exit(0)
    main()
    segmenter_c4()
