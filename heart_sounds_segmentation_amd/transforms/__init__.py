"""Mirror of ``hss.transforms`` (/root/reference/hss/transforms/__init__.py:1-8) for the hot path."""
from .synchrosqueeze import FSST

__all__ = ["FSST"]
