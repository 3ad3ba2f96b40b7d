"""Mirror of ``hss.moments`` (/root/reference/hss/moments/__init__.py:1-36): same names, argument
order and meaning; the arithmetic runs in the C-ABI library (hssfsst_update_mean/_variance)."""
from .._lib import lib as _lib


def update_mean(m: float, x: float, k: int) -> float:
    """Running mean after seeing ``x`` as the k-th value (hss/moments/__init__.py:16)."""
    return _lib().hssfsst_update_mean(float(m), float(x), int(k))


def update_variance(x: float, m: float, var: float, k: int) -> float:
    """Welford M2 accumulation; ``m`` is the mean BEFORE ``x`` (hss/moments/__init__.py:35-36).
    Like the reference it returns the un-divided sum of squares; the caller divides."""
    return _lib().hssfsst_update_variance(float(x), float(m), float(var), int(k))


__all__ = ["update_mean", "update_variance"]
