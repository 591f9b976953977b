"""Fused float containers (``modules``), their QAT (``qat``) and int8 (``quantized``) forms."""
from . import modules as _m

__all__ = ["SpconvReLUNd", "SpconvBnNd", "SpconvBnReLUNd", "SpconvAddReLUNd", "SpconvBnAddReLUNd",
           "_FusedSparseModule"]
globals().update({name: getattr(_m, name) for name in __all__})
