from .modules import (SpconvAddReLUNd, SpconvBnAddReLUNd, SpconvBnNd, SpconvBnReLUNd,  # noqa: F401
                      SpconvReLUNd, _FusedSparseModule)
