"""Container modules that combine several sparse tensors (reference ``spconv/pytorch/tables.py:25-92``).

``JoinTable`` concatenates channels, ``AddTable`` sums features of tensors that share one
coordinate set, ``AddTableMisaligned`` unions the coordinate sets first (hash based,
``functional.sparse_add_hash_based``), ``ConcatTable`` applies every child to the same input."""
from __future__ import annotations

from typing import List

import torch

from spconv_amd.pytorch import functional as F
from spconv_amd.pytorch.core import SparseConvTensor
from spconv_amd.pytorch.modules import SparseModule


def _check_aligned(tensors: List[SparseConvTensor], msg: str) -> SparseConvTensor:
    head = tensors[0]
    for t in tensors:
        assert t.spatial_shape == head.spatial_shape, msg
        assert t.batch_size == head.batch_size, msg
        assert t.features.shape[1] == head.features.shape[1], msg
        assert t.indices.shape[0] == head.indices.shape[0], msg
    return head


def _combined(tensors: List[SparseConvTensor], features: torch.Tensor) -> SparseConvTensor:
    head = tensors[0]
    out = SparseConvTensor(features, head.indices, head.spatial_shape, head.batch_size, head.grid,
                           head.voxel_num, head.indice_dict)
    # bookkeeping follows the second operand, as the reference does (tables.py:38-40)
    src = tensors[1] if len(tensors) > 1 else head
    out.benchmark_record = src.benchmark_record
    out.thrust_allocator = src.thrust_allocator
    out._timer = src._timer
    return out


class JoinTable(SparseModule):
    def forward(self, input: List[SparseConvTensor]):
        _check_aligned(input, "you can't use JoinTable in two sptensor with different indices.")
        return _combined(input, torch.cat([t.features for t in input], 1))

    def input_spatial_size(self, out_size):
        return out_size


class AddTable(SparseModule):
    def forward(self, input: List[SparseConvTensor]):
        _check_aligned(input, "you can't use AddTable in two sptensor with different indices. "
                              "use AddTableMisaligned instead.")
        return _combined(input, sum(t.features for t in input))

    def input_spatial_size(self, out_size):
        return out_size


class AddTableMisaligned(SparseModule):
    """Adds tensors of one shape but different coordinate sets.  The result carries a fresh
    coordinate set, so cached downsample rulebooks (SparseInverseConv) no longer apply."""

    def forward(self, input: List[SparseConvTensor]):
        return F.sparse_add_hash_based(*input)

    def input_spatial_size(self, out_size):
        return out_size


class ConcatTable(SparseModule):
    def forward(self, input):
        return [m(input) for m in self._modules.values()]

    def add(self, module):
        self._modules[str(len(self._modules))] = module
        return self

    def input_spatial_size(self, out_size):
        return self._modules["0"].input_spatial_size(out_size)
