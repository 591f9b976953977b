"""Fused float containers the quantization passes look for (reference
``quantization/intrinsic/modules.py:12-86``): ``conv (+ bn) (+ add) (+ relu)`` kept as the
original modules, executed in order; later replaced by one QAT or quantised module."""
from __future__ import annotations

import torch.ao.nn.intrinsic as nni
from torch.nn import BatchNorm1d, ReLU

from spconv_amd.pytorch.conv import SparseConvolution
from spconv_amd.pytorch.core import SparseConvTensor
from spconv_amd.pytorch.modules import is_spconv_module


def _check(mods, kinds):
    ok = len(mods) == len(kinds) and all(isinstance(m, k) for m, k in zip(mods, kinds))
    assert ok, "Incorrect types for input modules" + "".join(str(type(m)) for m in mods)


class _FusedSparseModule(nni._FusedModule):
    """Sequential over (sparse conv, dense layers...): dense layers see ``.features``."""

    def forward(self, input):
        for module in self._modules.values():
            if is_spconv_module(module):
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    input = input.replace_feature(module(input.features))
            else:
                input = module(input)
        return input


class SpconvReLUNd(_FusedSparseModule):
    def __init__(self, conv, relu):
        _check((conv, relu), (SparseConvolution, ReLU))
        super().__init__(conv, relu)


class SpconvBnNd(_FusedSparseModule):
    def __init__(self, conv, bn):
        _check((conv, bn), (SparseConvolution, BatchNorm1d))
        super().__init__(conv, bn)


class SpconvBnReLUNd(_FusedSparseModule):
    def __init__(self, conv, bn, relu):
        _check((conv, bn, relu), (SparseConvolution, BatchNorm1d, ReLU))
        super().__init__(conv, bn, relu)


class SpconvBnAddReLUNd(_FusedSparseModule):
    """relu(bn(conv(x)) + residual): the tail of a residual block."""

    def __init__(self, conv, bn, relu):
        _check((conv, bn, relu), (SparseConvolution, BatchNorm1d, ReLU))
        super().__init__(conv, bn, relu)

    def forward(self, input, add_input):
        conv, bn, relu = self[0], self[1], self[2]
        y = conv(input)
        return y.replace_feature(relu(bn(y.features) + add_input.features))


class SpconvAddReLUNd(_FusedSparseModule):
    """relu(conv(x) + residual) (bn already folded into the conv)."""

    def __init__(self, conv, relu):
        _check((conv, relu), (SparseConvolution, ReLU))
        super().__init__(conv, relu)

    def forward(self, input, add_input):
        conv, relu = self[0], self[1]
        y = conv(input)
        return y.replace_feature(relu(y.features + add_input.features))
