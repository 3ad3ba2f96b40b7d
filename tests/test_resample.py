"""Resample: host helper ``hssfsst_resample`` + the Python mirror of
``hss.transforms.Resample`` against (1) fixtures produced by the reference's own class in the build container
(tests/golden/resample.npz, made by tests/golden/make_golden.py) and (2) the numpy restatement in oracle/.
The first three tests read like the reference's test/test_transforms.py.  CPU only: nothing here needs a GPU."""
import os

import numpy as np
import pytest
import torch

from heart_sounds_segmentation_amd import synth
from heart_sounds_segmentation_amd.transforms import FSST, Resample
from heart_sounds_segmentation_amd.transforms.resample import resample_labels

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "resample.npz"))
CASES = sorted({k.split("__")[0] for k in GOLD.files if k.endswith("__x")})


@pytest.fixture
def resample_transform():
    return Resample(num=100)


@pytest.fixture
def input_tensor():
    return torch.tensor([1, 2, 3, 5])


def test_resample_output_type(resample_transform, input_tensor):          # test/test_transforms.py:17-19
    assert isinstance(resample_transform(input_tensor), torch.Tensor)


def test_resample_output_shape(resample_transform, input_tensor):         # test/test_transforms.py:22-24
    y = resample_transform(input_tensor)
    assert y.shape == (100,) and y.dtype == torch.float32 and y.device.type == "cpu"


def test_resample_different_input_sizes():                                # test/test_transforms.py:33-42
    f = Resample(num=50)
    assert f(torch.tensor([1, 2, 3])).shape == (50,)
    assert f(torch.tensor([1, 2, 3, 4, 5])).shape == (50,)


def test_resample_range_behaviour_is_the_references(resample_transform, input_tensor):
    """The reference's own test_resample_preserves_range (test/test_transforms.py:27-30) FAILS with the reference
    (Fourier resampling overshoots: min 0.72 < 1); a drop-in must overshoot identically."""
    y = resample_transform(input_tensor)
    ref = torch.from_numpy(GOLD["ref_test_input_100__y"])
    assert float(ref.min()) < 1.0 and float(ref.max()) > 5.0
    assert torch.allclose(y, ref, rtol=0, atol=2e-6)


@pytest.mark.parametrize("tag", CASES)
def test_golden_from_reference_class(tag):
    x, num, ref = GOLD[f"{tag}__x"], int(GOLD[f"{tag}__num"]), GOLD[f"{tag}__y"]
    got = Resample(num)(torch.from_numpy(x), torch.float64).numpy()
    assert got.shape == ref.shape
    # float64 fixtures: fp64 vs fp64; float32 fixtures: the reference ran scipy's single-precision FFT / cast to fp32
    tol = 1e-12 if ref.dtype == np.float64 else 1e-6
    assert np.abs(got - ref).max() <= tol * max(np.abs(ref).max(), 1.0)
    # default dtype, as the dataset calls it
    got32 = Resample(num)(torch.from_numpy(x))
    assert got32.dtype == torch.float32 and np.abs(got32.numpy() - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1.0)


def test_label_rule_matches_reference_dataset():
    """hss/datasets/heart_sounds.py:205-206: y = round(Resample(y)) - 1, int64."""
    y = torch.from_numpy(GOLD["labels__y"])
    t = Resample(int(GOLD["labels__num"]))
    out = resample_labels(y, t)
    assert out.dtype == torch.int64
    raw = GOLD["labels__raw"]
    robust = np.abs(raw - np.floor(raw) - 0.5) > 1e-9          # not sitting on a rounding tie
    assert robust.mean() > 0.99
    assert np.array_equal(out.numpy()[robust], GOLD["labels__out"][robust])


@pytest.mark.parametrize("n,num", [(35500, 8875), (35500, 17751), (2000, 2000), (2, 3), (3, 2), (1024, 1000), (997, 4001)])
def test_against_numpy_restatement(n, num):
    from oracle.resample_numpy import resample as oracle_resample
    x = np.random.default_rng(n + num).standard_normal(n)
    got = Resample(num)(torch.from_numpy(x), torch.float64).numpy()
    ref = oracle_resample(x, num)
    assert np.abs(got - ref).max() <= 1e-11 * max(np.abs(ref).max(), 1.0)


def test_identity_and_round_trip():
    x = torch.from_numpy(synth.pcg_windows(1, 600, seed=2)[0]).to(torch.float64)
    assert torch.allclose(Resample(600)(x, torch.float64), x, atol=1e-13)
    up = Resample(1800)(x, torch.float64)                      # band-limited interpolation is exactly invertible
    assert torch.allclose(Resample(600)(up, torch.float64), x, atol=1e-12)
    assert torch.allclose(up[::3], x, atol=1e-12)


def test_shapes_and_errors():
    col = torch.arange(10, dtype=torch.float32).reshape(10, 1)
    assert Resample(5)(col).shape == (5, 1)
    with pytest.raises(ValueError):
        Resample(5)(torch.zeros(4, 3))
    with pytest.raises(ValueError):
        Resample(0)(torch.zeros(4))
    with pytest.raises(ValueError):
        Resample(4)(torch.zeros(0))


def test_dataset_scan_finds_the_mirror_class():
    """heart_sounds.py:202-207 scans ``transform.transforms`` with isinstance(t, Resample)."""
    chain = [Resample(1000), "placeholder-for-FSST"]
    assert [isinstance(t, Resample) for t in chain] == [True, False]
    assert FSST.__module__.startswith("heart_sounds_segmentation_amd.transforms")
