from .modules import (SparseConv, SparseConvAddReLU, SparseConvBn, SparseConvBnAddReLU,  # noqa: F401
                      SparseConvBnReLU, SparseConvReLU)
