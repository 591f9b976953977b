"""QAT forms of the fused sparse-conv containers (see ``modules``)."""
from . import modules as _m

__all__ = ["SparseConv", "SparseConvReLU", "SparseConvAddReLU", "SparseConvBn", "SparseConvBnReLU",
           "SparseConvBnAddReLU"]
globals().update({name: getattr(_m, name) for name in __all__})
