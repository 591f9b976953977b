"""``spconv.pytorch.spatial`` (reference ``spconv/pytorch/spatial.py:28-45``)."""
import torch

from spconv_amd.pytorch.core import SparseConvTensor
from spconv_amd.pytorch.modules import SparseModule


class RemoveDuplicate(SparseModule):
    """Keeps one row per (batch, coordinate): the FIRST one in row order, which is also the row the
    rulebook's hash keeps for a duplicated coordinate (csrc/sparse/indices.py:1672).  The reference
    linearises the indices and takes ``torch.unique`` of the keys; so does this, with the row order
    made explicit."""

    def forward(self, x: SparseConvTensor) -> SparseConvTensor:
        inds = x.indices
        key = inds[:, 0].to(torch.int64)
        for d, size in enumerate(x.spatial_shape):
            key = key * int(size) + inds[:, d + 1].to(torch.int64)
        uniq, inverse = torch.unique(key, return_inverse=True)
        rows = torch.arange(inds.shape[0], device=inds.device)
        first = torch.full((uniq.shape[0],), inds.shape[0], dtype=rows.dtype, device=inds.device)
        first.scatter_reduce_(0, inverse, rows, reduce="amin")
        first = first.sort().values                      # keep the surviving rows in their order
        return SparseConvTensor(x.features[first], inds[first].contiguous(), x.spatial_shape,
                                x.batch_size, x.grid)
