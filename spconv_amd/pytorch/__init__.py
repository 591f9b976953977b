"""``import spconv_amd.pytorch as spconv`` -- the reference's ``spconv.pytorch`` namespace
(``spconv/pytorch/__init__.py:1-41``) for the convolution hot path."""
from spconv_amd.pytorch import conv, core, functional, hash, identity, modules, ops, pool, tables, utils
from spconv_amd.pytorch.conv import (SparseConv1d, SparseConv2d, SparseConv3d, SparseConv4d,
                                     SparseConvolution, SparseConvTranspose1d,
                                     SparseConvTranspose2d, SparseConvTranspose3d,
                                     SparseConvTranspose4d, SparseInverseConv1d,
                                     SparseInverseConv2d, SparseInverseConv3d,
                                     SparseInverseConv4d, SubMConv1d, SubMConv2d, SubMConv3d,
                                     SubMConv4d)
from spconv_amd.pytorch.core import ConvAlgo, SparseConvTensor
from spconv_amd.pytorch.identity import Identity
from spconv_amd.pytorch.tables import AddTable, AddTableMisaligned, ConcatTable, JoinTable
from spconv_amd.pytorch.pool import (SparseAvgPool1d, SparseAvgPool2d, SparseAvgPool3d,
                                     SparseGlobalAvgPool, SparseGlobalMaxPool, SparseMaxPool1d,
                                     SparseMaxPool2d, SparseMaxPool3d, SparseMaxPool4d)
from spconv_amd.pytorch.modules import (SparseBatchNorm, SparseIdentity, SparseModule, SparseReLU,
                                        SparseSequential, SparseSyncBatchNorm,
                                        assign_name_for_sparse_modules)


class ToDense(SparseModule):
    """convert SparseConvTensor to NCHW dense tensor."""

    def forward(self, x: SparseConvTensor):
        return x.dense()


class RemoveGrid(SparseModule):
    """remove pre-allocated grid buffer."""

    def forward(self, x: SparseConvTensor):
        x.grid = None
        return x
