from .conv import SparseConv  # noqa: F401
