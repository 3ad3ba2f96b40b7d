import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """The in-tree HIP library (built here by hipcc cross-compilation if needed)."""
    from heart_sounds_segmentation_amd import _lib
    _lib.build()
    return _lib.lib()


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


def pytest_sessionfinish(session, exitstatus):
    """The parity gate's tally of the session (tests/parity.py): fragile columns met, and how many of them differed."""
    try:
        from tests import parity
    except Exception:
        return
    t = parity.TALLY
    if not t["checks"]:
        return
    line = (f"parity tally: {t['checks']} signal checks, {t['columns']} columns, {t['fragile']} within {parity.FRAG_EPS:g} bins of a rounding tie, "
            f"{t['flipped']} of those differ from the oracle, {t['flipped_strict']} of them farther than {parity.FRAG_STRICT:g} from the tie")
    print("\n" + line)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_tally.txt"), "w") as fh:
            fh.write(line + "\n")
