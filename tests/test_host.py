"""CPU tests of the host-side mirror of the reference interface (hss.transforms.FSST,
hss.moments): constructor / attribute parity, error behaviour, fork/pickle safety, and that the
product never computes on the CPU."""
import inspect
import pickle

import numpy as np
import pytest
import torch

from heart_sounds_segmentation_amd import FSST, synth, transforms

KAISER = synth.kaiser_window(128, 0.5)
NO_GPU = not torch.cuda.is_available()


def test_constructor_signature_matches_reference():
    # /root/reference/hss/transforms/synchrosqueeze.py:13-21
    params = list(inspect.signature(FSST.__init__).parameters)
    assert params[:7] == ["self", "fs", "window", "abs", "stack", "truncate_freq", "dtype"]
    sig = inspect.signature(FSST.__init__).parameters
    assert sig["abs"].default is False and sig["stack"].default is False
    assert sig["truncate_freq"].default is None and sig["dtype"].default is torch.float32
    tf = FSST(1000, KAISER, truncate_freq=(25, 200), stack=True)
    assert tf.fs == 1000 and tf.window is KAISER and tf.abs is False and tf.stack is True
    assert tf.truncate_freq == (25, 200) and tf.dtype is torch.float32
    assert transforms.__all__ == ["Resample", "FSST"] and callable(tf)      # hss/transforms/__init__.py:5-8


def test_band_geometry_on_host(built_lib):
    assert FSST(1000, KAISER, truncate_freq=(25, 200)).band() == (4, 22)
    assert FSST(1000, KAISER).band() == (0, 65)
    assert FSST(1000, KAISER, truncate_freq=(1, 2)).band()[1] == 0


def test_truncate_valueerror_contract(built_lib):
    with pytest.raises(ValueError, match="truncate_freq must be set"):     # synchrosqueeze.py:104-105
        FSST(1000, KAISER)._truncate_frequencies(torch.zeros(65, 3), torch.zeros(65))
    s, f = FSST(1000, KAISER, truncate_freq=(25, 200))._truncate_frequencies(
        torch.arange(65.0)[:, None].repeat(1, 3), torch.arange(65.0) * 7.8125)
    assert s.shape == (22, 3) and f[0] == 31.25 and f[-1] == 195.3125


def test_pickle_drops_device_handles(built_lib):
    tf = FSST(1000, KAISER, truncate_freq=(25, 200), stack=True)
    tf._plans[("fake",)] = object()
    tf2 = pickle.loads(pickle.dumps(tf))
    assert tf2._plans == {} and tf2.truncate_freq == (25, 200) and np.array_equal(tf2.window, KAISER)


@pytest.mark.skipif(not NO_GPU, reason="checks the no-device failure mode")
def test_no_cpu_fallback(built_lib):
    tf = FSST(1000, KAISER, truncate_freq=(25, 200), stack=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tf(torch.zeros(2000))
    with pytest.raises(RuntimeError, match="no CPU"):
        FSST(1000, KAISER, device="cpu").batch(torch.zeros(2, 100))


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from heart_sounds_segmentation_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_product_never_imports_oracle():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "heart_sounds_segmentation_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in text and "from oracle" not in text, fn
                assert "hss_oracle" not in text, fn


def test_frame_signal_golden_contract():
    """Window set fed to the path (hss/utils/preprocess.py:40-56 semantics pinned by the fixture):
    L = floor((T-n)/stride) frames, or ONE frame x[:n] when L <= 0."""
    from heart_sounds_segmentation_amd import framing
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "frame_signal.npz"))
    for T in (35000, 35500, 4000, 3000, 2500, 2000, 1500):
        starts, lens = framing.frame_starts(T, 1000, 2000)
        assert np.array_equal(starts, g[f"T{T}__starts"]), T
        assert np.array_equal(lens, g[f"T{T}__lens"]), T
    x = torch.arange(35500, dtype=torch.float32)
    F = framing.frame_batch(x, 1000, 2000)
    assert F.shape == (33, 2000) and F[5, 0] == 5000 and F[32, -1] == 33999


def test_consumer_matches_reference_segmenter_golden():
    """tests/golden/segmenter.npz was produced by the reference's HeartSoundSegmenter
    (hss/model/segmenter.py:5-87): same state_dict keys load, same output."""
    import os
    from heart_sounds_segmentation_amd.consumer import SegmenterHead
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "segmenter.npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd__")}
    head = SegmenterHead(44, 12, 3, h0=torch.from_numpy(g["h0"]), c0=torch.from_numpy(g["c0"]))
    missing, unexpected = head.load_state_dict(sd, strict=True), None
    head.eval()
    with torch.no_grad():
        y = head(torch.from_numpy(g["x"]))
    assert y.shape == (3, 40, 4)
    assert np.abs(y.numpy() - g["y"]).max() < 1e-5


def test_consumer_c4_size_matches_reference_pipeline(oracle_mod):
    """BASELINE config C4 at its real size on the CPU: tests/golden/segmenter_c4.npz holds the output of the
    reference's own pipeline (reference FSST wrapper over the oracle core -> HeartSoundSegmenter(44, batch 50,
    hidden 240), hss/model/segmenter.py:20-87, main.py:170,221).  The weights are replayed from the seed (checksum
    pinned); the features come from the oracle's C epilogue."""
    import os
    from heart_sounds_segmentation_amd import synth
    from heart_sounds_segmentation_amd.consumer import SegmenterHead
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "segmenter_c4.npz"))
    head = SegmenterHead.seeded_like_reference(int(g["seed"])).eval()
    assert head.checksum() == g["sha256"].tobytes(), "weight replay differs from the reference-made fixture"
    X = synth.pcg_windows(50, 2000, seed=int(g["window_seed"]))
    feats = oracle_mod.features(X, 1000, synth.kaiser_window(128, 0.5), (25, 200), "stack", nthreads=8)
    assert np.abs(feats[:, ::250, :] - g["feat_probe"]).max() < 5e-6          # the reference wrapper's features
    with torch.no_grad():
        y = head(torch.from_numpy(feats))
    assert y.shape == (50, 2000, 4)
    assert np.abs(y.numpy() - g["y"]).max() < 2e-4                              # log-probabilities, absolute


def test_csv_ingest_matches_pandas(built_lib, tmp_path):
    """ingest.load_file == the reference's _load_file recipe (heart_sounds.py:193-197) on a synthetic
    file of the corpus format."""
    import pandas as pd
    from heart_sounds_segmentation_amd import ingest, _lib
    rng = np.random.default_rng(5)
    T = 5000
    sig = rng.standard_normal(T) * 0.3
    lab = rng.integers(1, 5, T)
    path = tmp_path / "a0001"
    with open(str(path) + ".csv", "w") as fh:
        fh.write("Signals,Labels\n")
        for i, (s, l) in enumerate(zip(sig, lab)):
            fh.write(f"{float(s)!r},{int(l)}\n" if i % 2 else f"{float(s):.9e},{int(l)}\n")
    df = pd.read_csv(str(path) + ".csv", skiprows=1, names=["Signals", "Labels"])
    x_ref = torch.tensor(df.loc[:, "Signals"].to_numpy(), dtype=torch.float32)
    y_ref = torch.tensor(df.loc[:, "Labels"].to_numpy(), dtype=torch.int64)
    x, y = ingest.load_file(str(path))
    assert x.dtype == torch.float32 and y.dtype == torch.int64 and x.shape == (T,)
    assert torch.equal(x, x_ref) and torch.equal(y, y_ref)
    with pytest.raises(ValueError):
        ingest.parse_csv_bytes(b"Signals,Labels\n0.5;1\n")
    assert ingest.parse_csv_bytes(b"Signals,Labels\n")[0].numel() == 0
    x2, y2 = ingest.parse_csv_bytes(b"Signals,Labels\r\n1.5,2.0\r\n\r\n-2e-3,4\n")
    assert x2.tolist() == [1.5, -0.0020000000949949026] and y2.tolist() == [2, 4]


def test_graft_entry_build_runs():
    """The driver's "does it build" check: __graft_entry__.build() compiles the library and the oracle and verifies
    the ABI version against the header (a hard-coded number once broke it after an ABI bump)."""
    import __graft_entry__ as g
    g.build()


def test_corpus_builder_groups_recordings_like_the_reference_loop():
    """SURVEY section 8f row 1 on the CPU: corpus.build_features lays recordings back to back and hands ALL frames of a
    group to one FSST.frames call; with a transform stand-in that returns the frames themselves, the items must be exactly
    what the reference's loop yields (heart_sounds.py:155-169: skip < frame_len, frame_signal's L = floor((T-n)/stride)
    frames -- one fewer than fit --, a single frame x[:n] when L <= 0, labels y - 1 framed alike), whatever the group size,
    and a rank's share is a contiguous block of recordings."""
    from heart_sounds_segmentation_amd.corpus import build_features
    from heart_sounds_segmentation_amd.framing import frame_batch

    class Identity:                                        # duck-typed FSST: features = the frame, as a (n, 1) block
        calls = 0

        def frames(self, x, starts, n, out=None):
            Identity.calls += 1
            return torch.stack([x[int(s):int(s) + n] for s in starts]).unsqueeze(-1)

    rng = np.random.default_rng(4)
    lens = (35000, 2500, 1999, 12345, 4000, 2000, 3000)
    recs = [(torch.from_numpy(rng.standard_normal(L).astype(np.float32)), torch.from_numpy(rng.integers(1, 5, L))) for L in lens]
    want = []
    for x, y in recs:                                      # the reference loop, restated with the golden-pinned framing
        if x.shape[0] < 2000:
            continue
        for fx, fy in zip(frame_batch(x, 1000, 2000), frame_batch(y - 1, 1000, 2000)):
            want.append((fx, fy))
    assert len(want) == 33 + 1 + 10 + 2 + 1 + 1
    for wpl in (1, 40, 4096):
        Identity.calls = 0
        got = build_features(recs, Identity(), device="cpu", windows_per_launch=wpl)
        assert len(got) == len(want)
        assert Identity.calls == {1: 6, 40: 2, 4096: 1}[wpl]       # (groups close at >= wpl frames: 33+1+10 | 2+1+1)
        for (gx, gy), (wx, wy) in zip(got, want):
            assert torch.equal(gx[:, 0], wx) and torch.equal(gy, wy) and gy.dtype == torch.int64
    parts = [build_features(recs, Identity(), device="cpu", rank=r, world=3) for r in range(3)]
    flat = [it for p in parts for it in p]
    assert len(flat) == len(want) and all(torch.equal(a[0][:, 0], b[0]) for a, b in zip(flat, want))


def test_pack_recordings_equals_python_framing(built_lib):
    """hssfsst_pack_recordings (the corpus builder's host side, one native call per group of recordings): the staging
    buffer is the recordings back to back and the frame starts are those of frame_signal
    (/root/reference/hss/utils/preprocess.py:30-58 as restated by framing.frame_starts), for lengths at and around the
    frame boundaries; capacity errors are reported, not overrun."""
    import ctypes
    from heart_sounds_segmentation_amd.framing import frame_starts
    recs = [torch.randn(T) for T in (35500, 2000, 2999, 3000, 3001, 4001, 35000, 2000)]
    ptrs = np.asarray([x.data_ptr() for x in recs], dtype=np.uint64)
    lens = np.asarray([x.numel() for x in recs], dtype=np.int64)
    stage = torch.empty(int(lens.sum()))
    starts = torch.empty(256, dtype=torch.int64)
    vp = ctypes.c_void_p
    for threads in (1, 3, 0):
        stage.zero_()
        nf = built_lib.hssfsst_pack_recordings(vp(ptrs.ctypes.data), vp(lens.ctypes.data), len(recs), 1000, 2000, vp(stage.data_ptr()),
                                               stage.numel(), vp(starts.data_ptr()), starts.numel(), threads)
        want, base = [], 0
        for x in recs:
            want.append(frame_starts(x.numel(), 1000, 2000)[0] + base)
            base += x.numel()
        want = np.concatenate(want)
        assert nf == len(want) and np.array_equal(starts[:nf].numpy(), want)
        assert torch.equal(stage, torch.cat(recs))
    assert built_lib.hssfsst_pack_recordings(vp(ptrs.ctypes.data), vp(lens.ctypes.data), len(recs), 1000, 2000, vp(stage.data_ptr()),
                                             stage.numel() - 1, vp(starts.data_ptr()), starts.numel(), 1) < 0
    assert built_lib.hssfsst_pack_recordings(vp(ptrs.ctypes.data), vp(lens.ctypes.data), len(recs), 1000, 2000, vp(stage.data_ptr()),
                                             stage.numel(), vp(starts.data_ptr()), 10, 1) < 0


def test_team_kernel_fixed_registers_are_nobodys_else(tmp_path):
    """fsst_team16_kernel keeps its held images in v104..v127 through inline assembly and tells the register allocator to stay
    below (amdgpu_num_vgpr: fsst_team16.hpp).  Checked on the listing: inside the team kernels the only instructions that name
    a register >= v104 are the two accessors (v_pk_mul_f32 into a pair, v_pk_add_f32 out of one), and the code objects reserve
    128 registers.  (A compiler that stopped honouring the limit would silently corrupt features.)"""
    import os
    import re
    import shutil
    import subprocess
    import pytest
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "dev.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-DHSS_DEV", "-DHSS_DEV_ONLY128", "--cuda-device-only", "-S",
                        "-o", str(out), os.path.join(root, "heart_sounds_segmentation_amd", "csrc", "hssfsst.hip")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text().split("\n")
    starts = [i for i, ln in enumerate(text) if re.match(r"^_ZN7hssfsst18fsst_team16_kernel.*:", ln)]
    assert len(starts) >= 2, "team kernels not found in the listing"
    put = re.compile(r"^\s*v_pk_mul_f32 v\[(\d+):(\d+)\], v\[\d+:\d+\], v\[\d+:\d+\]\s*$")
    get = re.compile(r"^\s*v_pk_add_f32 v\[\d+:\d+\], v\[(\d+):(\d+)\], v\[\d+:\d+\] op_sel")
    for st in starts:
        seen = set()
        for ln in text[st:]:
            if ln.startswith(".Lfunc_end"):
                break
            code = ln.split(";")[0]
            regs = [int(a) for a in re.findall(r"\bv(\d+)\b", code)] + [int(b) for _, b in re.findall(r"\bv\[(\d+):(\d+)\]", code)]
            if not regs or max(regs) < 104:
                continue
            m = put.match(code) or get.match(code)
            assert m and int(m.group(1)) >= 104 and int(m.group(2)) == int(m.group(1)) + 1, f"register >= v104 outside the accessors: {ln.strip()}"
            others = [int(b) for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", code)]
            assert sum(1 for b in others if b >= 104) == 1, ln
            seen.add(int(m.group(1)))
        assert seen == set(range(104, 128, 2)), sorted(seen)
    names = [i for i, ln in enumerate(text) if ".amdhsa_kernel _ZN7hssfsst18fsst_team16_kernel" in ln]
    assert len(names) >= 2
    for i in names:
        blk = "\n".join(text[i:i + 60])
        assert re.search(r"\.amdhsa_next_free_vgpr 128\b", blk), blk[:400]


def test_shipped_code_object_keeps_the_fixed_registers():
    """the check of `test_team_kernel_fixed_registers_are_nobodys_else` on the SHIPPED library (the code object inside
    libhssfsst.so, disassembled), plus: no AGPR, no call, 128 registers reserved.  __graft_entry__.build() runs the same check."""
    import os
    import pytest
    from heart_sounds_segmentation_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump")
    out = _lib.check_code_object()
    assert out["kernels"] >= 2 and out["metadata_checked"] >= 2 and out["accessor_instructions"] >= 100, out


def test_lent_buffer_goes_back_when_the_last_view_dies():
    """The drop-in call's result is a tensor over a pinned pool buffer the library LENT (hssfsst_exec_pinned): the ctypes array the tensor is made of
    hands the buffer back from its ``__del__`` -- once, with the buffer's address, and only when the last view of the result is gone."""
    import ctypes
    import os

    from heart_sounds_segmentation_amd.transforms import synchrosqueeze as sq

    backing = np.arange(2000 * 44, dtype=np.float32)
    released = []

    class FakeL:
        @staticmethod
        def hssfsst_pinned_release(handle, ptr):
            released.append(ptr.value)
            return 0

    class FakePlan:
        handle = ctypes.c_void_p(1234)
        pid = os.getpid()
        _L = FakeL

    buf = sq._lent_type(backing.size).from_address(backing.ctypes.data)
    buf._plan = FakePlan
    y = torch.from_numpy(np.frombuffer(buf, dtype=np.float32).reshape(2000, 44))
    del buf
    assert y.shape == (2000, 44) and float(y[1, 0]) == 44.0 and released == []
    v = y[10:12]
    del y
    assert released == []                                # a view keeps the buffer on loan
    assert float(v[0, 1]) == 441.0
    del v
    assert released == [backing.ctypes.data]
    assert sq._lent_type(backing.size) is sq._lent_type(backing.size)      # one array type per result size
