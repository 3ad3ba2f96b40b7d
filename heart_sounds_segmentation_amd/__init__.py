"""heart_sounds_segmentation_amd -- MI355X-native FSST feature path (drop-in for the reference's
``hss.transforms.FSST`` + ``hss.moments``).  See DESIGN.md and INTEGRATION.md."""
from . import moments, transforms  # noqa: F401
from .transforms import FSST, Resample  # noqa: F401

__all__ = ["FSST", "Resample", "moments", "transforms"]
