"""CPU oracle for the FSST feature path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product (``heart_sounds_segmentation_amd``) never does.  Parity status: **parity
unpinned** for the FSST core (the reference's native ``ssq``/``libssq`` dependency is absent; see
``oracle/fsst_oracle.c`` header and DESIGN.md); the wrapper epilogue is pinned by tests/golden.
"""
from .binding import (  # noqa: F401
    band,
    build,
    dtwin,
    features,
    fsst,
    lib_path,
    max_threads,
    update_mean,
    update_variance,
)
