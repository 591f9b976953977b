from .conv_relu import SparseConvAddReLU, SparseConvReLU  # noqa: F401
