"""Fuser methods for conv + bn (+ relu) (+ residual add) patterns (reference
``quantization/fuse_mapping.py:10-99``): eval mode folds the BatchNorm into the conv weights,
QAT mode keeps the modules in an intrinsic container that ``prepare_qat`` swaps for the QAT op."""
from __future__ import annotations

from spconv_amd.pytorch.conv import SparseConvolution
from spconv_amd.pytorch.quantization import intrinsic as snni
from spconv_amd.pytorch.quantization.utils import fuse_spconv_bn_eval


def _check_qat_bn(conv, bn):
    assert isinstance(conv, SparseConvolution), f"Cannot fuse train modules: {(conv, bn)}"
    assert bn.num_features == conv.out_channels, "Output channel of the conv must match num_features of BatchNorm"
    assert bn.affine, "Only support fusing BatchNorm with affine set to True"
    assert bn.track_running_stats, "Only support fusing BatchNorm with tracking_running_stats set to True"


def fuse_conv_bn(is_qat, conv, bn, is_add_fuse: bool = False):
    assert conv.training == bn.training, "Conv and BN both must be in the same mode (train or eval)."
    if not is_qat:
        return fuse_spconv_bn_eval(conv, bn)
    _check_qat_bn(conv, bn)
    return (snni.SpconvAddReLUNd if is_add_fuse else snni.SpconvBnNd)(conv, bn)


def fuse_conv_bn_relu(is_qat, conv, bn, relu, is_add_fuse: bool = False):
    assert conv.training == bn.training == relu.training, \
        "Conv and BN both must be in the same mode (train or eval)."
    if is_qat:
        _check_qat_bn(conv, bn)
        return (snni.SpconvBnAddReLUNd if is_add_fuse else snni.SpconvBnReLUNd)(conv, bn, relu)
    if not isinstance(conv, SparseConvolution):
        raise NotImplementedError(f"Cannot fuse eval modules: {(conv, bn, relu)}")
    return (snni.SpconvAddReLUNd if is_add_fuse else snni.SpconvReLUNd)(fuse_spconv_bn_eval(conv, bn), relu)


def fuse_conv_bn_add_relu(is_qat, relu, add_pattern):
    """``relu(add(bn(conv(x)), y))`` matched in the fx complex-pattern format."""
    _, bn_pattern, _ = add_pattern
    bn, conv = bn_pattern
    return fuse_conv_bn_relu(is_qat, conv, bn, relu, True)
