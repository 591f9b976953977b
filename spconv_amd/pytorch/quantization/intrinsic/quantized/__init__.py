"""int8 conv + ReLU / conv + residual add + ReLU modules (one kernel launch each)."""
from . import conv_relu as _m

__all__ = list(_m.__all__)
globals().update({name: getattr(_m, name) for name in __all__})
