"""Statically quantised sparse convolution (``conv``) and its reference form (``reference``)."""
from . import conv as _m

__all__ = ["SparseConv"]
SparseConv = _m.SparseConv
