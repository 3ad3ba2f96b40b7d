"""Shared parity gate (SURVEY.md section 8d "Parity gate"; BASELINE.json: features within 1e-4 rel).

Per signal:  max|out - ref| <= TOL * max|ref|  and  ||out - ref||_2 <= TOL * ||ref||_2, evaluated on
the time columns that are NOT rounding-fragile.  A column is fragile when some source cell's
reassignment coordinate lies within FRAG_EPS of a rounding tie in the fp64 oracle (SURVEY section 7
"discontinuous rounding"; round 1 needed 1e-3 here and budgets of 3-25 % of the columns, since round 2
the GPU path resolves such cells in float64 itself).  Fragile columns are counted, never silently dropped: those of them
that DIFFER from the oracle by more than TOL (a cell rounded the other way) must lie within FRAG_STRICT of the tie (a tie at
float64 resolution: a different summation order may round it the other way) and are at most two per signal --
a fragile column that agrees is just a column: tools/fuzz_parity.py seed 13 drew a signal with three of them, all within
8e-8 of the oracle -- and inside them the error must still be explainable by a moved cell (|err| bounded by twice the
signal's largest feature).
"""
import numpy as np

TOL = 1e-4          # the tolerance north_star states (fp32, relative to the signal's max feature)
FRAG_EPS = 1e-7     # distance from a rounding tie below which a column is "fragile": the kernels decide every
                    # rounding that float32 cannot call in float64 (fsst_mfma128.hpp "Rounding ties"), so only ties
                    # at float64 resolution remain (a different summation order may round them the other way)
FRAG_BUDGET = 0.0   # measured (profiles/r04_parity_tally.txt: 3.4 M columns of the GPU suite, 114 fragile, 0 differ): no allowance
FRAG_STRICT = 1e-9  # a fragile column farther than this from its tie must NOT flip at all: only ties at float64 resolution may

# running tally over a test session (tests/conftest.py prints it and writes gpurun_out/parity_tally.txt): how many fragile
# columns there were, how many of them differed from the oracle, how many of those lay farther than FRAG_STRICT from the tie
TALLY = {"checks": 0, "columns": 0, "fragile": 0, "flipped": 0, "flipped_strict": 0}


def check(out, ref, halfdist, time_axis, tol=TOL, frag_eps=FRAG_EPS, frag_budget=FRAG_BUDGET, what=""):
    """out/ref: one signal's features (real or complex ndarray); halfdist: (n,) from the oracle;
    time_axis: which axis of out is time.  Returns a dict of measurements; raises AssertionError."""
    out = np.asarray(out)
    ref = np.asarray(ref)
    assert out.shape == ref.shape, f"{what}: shape {out.shape} != {ref.shape}"
    assert out.dtype == ref.dtype, f"{what}: dtype {out.dtype} != {ref.dtype}"
    o = np.moveaxis(out, time_axis, 0).reshape(out.shape[time_axis], -1)
    r = np.moveaxis(ref, time_axis, 0).reshape(ref.shape[time_axis], -1)
    n = o.shape[0]
    fragile = np.asarray(halfdist) < frag_eps
    robust = ~fragile
    scale = float(np.abs(r).max()) if r.size else 0.0
    err = np.abs(o - r)
    assert np.isfinite(o[robust]).all() or not np.isfinite(r[robust]).all(), f"{what}: non-finite output"
    max_err = float(err[robust].max()) if robust.any() and err.shape[1] else 0.0
    l2 = float(np.linalg.norm((o - r)[robust])) if robust.any() else 0.0
    l2ref = float(np.linalg.norm(r[robust])) if robust.any() else 0.0
    nfrag = int(fragile.sum())
    # fragile columns that actually differ (a cell rounded the other way); a fragile column that agrees is just a column
    nflip = int((err[fragile].max(axis=1) > tol * scale).sum()) if nfrag and err.shape[1] else 0
    strict = fragile & (np.asarray(halfdist) > FRAG_STRICT)
    nstrict = int((err[strict].max(axis=1) > tol * scale).sum()) if strict.any() and err.shape[1] else 0
    TALLY["checks"] += 1; TALLY["columns"] += n; TALLY["fragile"] += nfrag; TALLY["flipped"] += nflip; TALLY["flipped_strict"] += nstrict
    res = dict(max_err=max_err, scale=scale, rel=max_err / scale if scale else 0.0,
               rel_l2=l2 / l2ref if l2ref else 0.0, fragile=nfrag, flipped=nflip, flipped_strict=nstrict, n=n)
    assert max_err <= tol * scale, f"{what}: max err {max_err:.3e} > {tol:g} * {scale:.3e} ({res})"
    assert l2 <= tol * l2ref, f"{what}: rel L2 {res['rel_l2']:.3e} > {tol:g} ({res})"
    # (only a tie at float64 resolution -- within FRAG_STRICT -- may round the other way: at most two such columns per signal)
    assert nflip <= max(2, frag_budget * n), f"{what}: {nflip}/{n} fragile columns differ, over the budget ({nfrag} fragile)"
    assert nstrict == 0, f"{what}: {nstrict} columns {FRAG_STRICT:g} .. {frag_eps:g} bins from a rounding tie differ from the oracle ({res})"
    if nfrag and err.shape[1]:
        # a flipped cell moves at most its own magnitude between two rows
        # a flipped cell moves at most one cell's magnitude (bounded by ~the signal's max) between rows
        assert float(err[fragile].max()) <= 2.0 * scale, f"{what}: fragile-column error not explainable"
    return res
