"""Mirror of ``hss.transforms`` (/root/reference/hss/transforms/__init__.py:1-8)."""
from .resample import Resample
from .synchrosqueeze import FSST

__all__ = ["Resample", "FSST"]
