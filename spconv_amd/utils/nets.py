"""The two whole networks BASELINE.json's configs name, built from this package's modules.

* ``downsample_chain`` -- config 3: SparseConv3d k3 s2 p1, 16 -> 32 -> 64 -> 128.
* ``second_backbone``  -- config 4: the SECOND / VoxelNet middle extractor (OpenPCDet's
  ``VoxelBackBone8x`` layout as SURVEY.md section 8d records it; the reference itself ships no
  SECOND model, its nearest in-tree nets are test/test_multi_impl.py:35-158 and
  benchmark/basic.py:16-146): SubM(C_in->16), SubM16, then three [SparseConv s2 + 2 x SubM] stages
  16->32->64->64 (the last with padding (0,1,1)) and a (3,1,1)/(2,1,1) SparseConv 64->128, every
  conv followed by BatchNorm1d + ReLU; SubM layers of one stage share an ``indice_key``.

Used by ``bench.py --config 3|4``, ``tools/netbench.py`` and the parity tests, so that all of
them time / check the same network.
"""
from __future__ import annotations

from torch import nn

import spconv_amd.pytorch as spconv

SECOND_SHAPE = [41, 1600, 1408]          # z, y, x: KITTI voxel grid of SECOND (0.05 m x 0.05 m x 0.1 m)


def _subm(cin, cout, key, norm):
    layers = [spconv.SubMConv3d(cin, cout, 3, bias=False, indice_key=key)]
    if norm:
        layers.append(nn.BatchNorm1d(cout, eps=1e-3, momentum=0.01))
    layers.append(nn.ReLU())
    return layers


def _down(cin, cout, key, k=3, s=2, p=1, norm=True):
    layers = [spconv.SparseConv3d(cin, cout, k, s, p, bias=False, indice_key=key)]
    if norm:
        layers.append(nn.BatchNorm1d(cout, eps=1e-3, momentum=0.01))
    layers.append(nn.ReLU())
    return layers


def second_backbone(cin: int = 4, norm: bool = True) -> spconv.SparseSequential:
    return spconv.SparseSequential(
        *_subm(cin, 16, "subm1", norm), *_subm(16, 16, "subm1", norm),
        *_down(16, 32, "spconv2", norm=norm), *_subm(32, 32, "subm2", norm), *_subm(32, 32, "subm2", norm),
        *_down(32, 64, "spconv3", norm=norm), *_subm(64, 64, "subm3", norm), *_subm(64, 64, "subm3", norm),
        *_down(64, 64, "spconv4", 3, 2, (0, 1, 1), norm=norm), *_subm(64, 64, "subm4", norm),
        *_subm(64, 64, "subm4", norm),
        *_down(64, 128, "spconv_down2", (3, 1, 1), (2, 1, 1), 0, norm=norm))


def downsample_chain(widths=(16, 32, 64, 128)) -> spconv.SparseSequential:
    return spconv.SparseSequential(*[spconv.SparseConv3d(a, b, 3, 2, 1, bias=False)
                                     for a, b in zip(widths[:-1], widths[1:])])


def conv_layers(net: nn.Module):
    """The sparse convolution modules of a network, in execution order of a Sequential."""
    from spconv_amd.pytorch.conv import SparseConvolution
    return [m for m in net.modules() if isinstance(m, SparseConvolution)]
