"""byolo -- Python host side of libbyolo.so, the MI355X-native Bayesian-YOLOv3 inference path.

Importing this package loads the HIP library; it raises ImportError if the library is missing
(there is no CPU / eager fallback)."""
from . import _lib
from ._lib import (ByoloError, DET_STANDARD, DET_ALEATORIC, DET_EPISTEMIC, NMS_AGNOSTIC, NMS_TWO_CLASS,
                   NORM_BN, NORM_DROPOUT, LIB_PATH)
from .engine import Engine

__all__ = ["Engine", "ByoloError", "DET_STANDARD", "DET_ALEATORIC", "DET_EPISTEMIC", "NMS_AGNOSTIC",
           "NMS_TWO_CLASS", "NORM_BN", "NORM_DROPOUT", "LIB_PATH"]
