#!/usr/bin/env python3
"""ONE COMMAND that turns "parity unpinned" into pinned, on any machine where the reference's native core is installed:

    python tools/pin_core.py            # needs `import ssq` (ssq 0.1.0 / libssq 0.1.0, channel https://prefix.dev/ssq,
                                        # /root/reference/pyproject.toml:26,42; e.g. inside the reference's pixi env)

It calls the REAL ``ssq.fsst(x, fs, window)`` -- the call of /root/reference/hss/transforms/synchrosqueeze.py:48 -- on
the inputs the committed fixtures already use (the seven wrapper cases of tests/golden/fsst_wrapper.npz, regenerated
from their seeds) plus the signals of tests/known_answers.py and two PCG / noise windows, and writes
``tests/golden/core_ssq.npz``: for every case ``<tag>__x``, ``__fs``, ``__window`` and the core's raw outputs
``__s`` (complex128 (nf, nt)), ``__f``, ``__t`` exactly as returned (no cast, no truncation).

Once that file is committed, tests/test_oracle.py::test_oracle_matches_real_ssq_core (CPU) and
tests/test_gpu_parity.py::test_hip_matches_real_ssq_core (-m gpu) compare the C oracle and the HIP path with it and stop
skipping.  Nothing of the reference is copied: the file holds inputs and numeric outputs only.

The build container and the GPU boxes have no ``ssq``: there this script exits with code 3.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "core_ssq.npz")


def cases():
    """(tag, x, fs, window) -- inputs only; the same seeds as tests/golden/make_golden.py where a case has a twin there."""
    from scipy.signal import get_window

    from heart_sounds_segmentation_amd import synth
    kaiser = synth.kaiser_window(128, 0.5)
    hann64 = get_window("hann", 64, fftbins=False)
    rng = np.random.default_rng(7)
    out = [("A_canonical", synth.pcg_windows(1, 2000, seed=11)[0], 1000.0, kaiser),
           ("B_noise600", rng.standard_normal(600).astype(np.float32), 1000.0, kaiser)]
    xc = rng.standard_normal(400).astype(np.float32)
    out += [("C_noise400", xc, 1000.0, kaiser),
            ("E_noise256", rng.standard_normal(256).astype(np.float32), 1000.0, kaiser),
            ("F_hann64_f64", rng.standard_normal(500), 2000.0, hann64),
            ("G_noise300", rng.standard_normal(300).astype(np.float32), 1000.0, kaiser)]
    # the closed-form signals of tests/known_answers.py (each pins one assumption of SURVEY appendix A)
    n = 600
    imp = np.zeros(n); imp[300] = 1.0
    out.append(("K_impulse_kaiser4", imp, 1000.0, np.kaiser(128, 4.0)))
    out.append(("K_impulse_odd", imp, 1000.0, np.kaiser(127, 4.0)))
    out.append(("K_constant", np.full(500, 2.5), 1000.0, np.kaiser(128, 3.0)))
    t = np.arange(1200) / 1000.0
    out.append(("K_offbin_tone_hann", np.cos(2 * np.pi * 16.3 * 1000.0 / 128 * t), 1000.0, get_window("hann", 128, fftbins=False)))
    t2 = np.arange(2000) / 1000.0
    out.append(("K_chirp_gauss", np.cos(2 * np.pi * (60.0 * t2 + 0.5 * 80.0 * t2 ** 2)), 1000.0,
                get_window(("gaussian", 16.0), 128, fftbins=False)))
    # the benchmark's own inputs and the worst case for the reassignment spread
    out.append(("P_pcg", synth.pcg_windows(2, 2000, seed=123)[1], 1000.0, kaiser))
    out.append(("N_noise", synth.noise_windows(1, 2000, seed=5)[0], 1000.0, kaiser))
    # other window lengths the plan accepts (a12: ANY window array)
    out.append(("W_kaiser256", synth.noise_windows(1, 1000, seed=6)[0], 1000.0, np.kaiser(256, 0.5)))
    out.append(("W_hamming100", synth.noise_windows(1, 700, seed=8)[0], 1000.0, get_window("hamming", 100, fftbins=False)))
    return out


def main():
    try:
        import ssq                                        # the reference's native core
    except ImportError as e:
        print(f"pin_core: `import ssq` failed ({e}): run this where the reference's environment is installed", file=sys.stderr)
        return 3
    blob = {}
    for tag, x, fs, w in cases():
        s, f, t = ssq.fsst(np.ascontiguousarray(x), fs, np.ascontiguousarray(w, dtype=np.float64))     # synchrosqueeze.py:48
        blob[f"{tag}__x"] = np.asarray(x)
        blob[f"{tag}__fs"] = np.float64(fs)
        blob[f"{tag}__window"] = np.asarray(w, dtype=np.float64)
        blob[f"{tag}__s"] = np.asarray(s)
        blob[f"{tag}__f"] = np.asarray(f)
        blob[f"{tag}__t"] = np.asarray(t)
        print(tag, np.asarray(x).shape, "->", np.asarray(s).shape, np.asarray(s).dtype)
    blob["__ssq_version"] = np.bytes_(getattr(ssq, "__version__", "unknown"))
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT)
    return 0


if __name__ == "__main__":
    sys.exit(main())
