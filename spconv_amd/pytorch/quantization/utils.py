"""Folding BatchNorm / activations into a sparse convolution for inference
(reference ``spconv/pytorch/quantization/utils.py:5-53``)."""
from __future__ import annotations

import copy

import torch

from spconv_amd.pytorch.ops import Activation


def fuse_spconv_bn_weights(conv_w, conv_b, bn_rm, bn_rv, bn_eps, bn_w, bn_b):
    """KRSC weight ``[K, *ksize, C]`` and bias of ``bn(conv(x))`` as one convolution:
    ``w' = w * gamma / sqrt(var + eps)`` per output channel (dim 0),
    ``b' = (b - mean) * gamma / sqrt(var + eps) + beta``."""
    if conv_b is None:
        conv_b = torch.zeros_like(bn_rm)
    if bn_w is None:
        bn_w = torch.ones_like(bn_rm)
    if bn_b is None:
        bn_b = torch.zeros_like(bn_rm)
    gain = bn_w * torch.rsqrt(bn_rv + bn_eps)
    w = (conv_w * gain.reshape([-1] + [1] * (conv_w.ndim - 1)).to(conv_w.dtype)).contiguous()
    b = (conv_b - bn_rm) * gain + bn_b
    return torch.nn.Parameter(w), torch.nn.Parameter(b.to(conv_w.dtype))


def fuse_spconv_bn_eval(conv, bn):
    """A copy of ``conv`` with ``bn`` folded in: C(x) == bn(conv(x)) in eval mode."""
    assert not (conv.training or bn.training), "Fusion only for eval!"
    fused = copy.deepcopy(conv)
    fused.weight, fused.bias = fuse_spconv_bn_weights(fused.weight, fused.bias, bn.running_mean,
                                                      bn.running_var, bn.eps, bn.weight, bn.bias)
    return fused


def fuse_spconv_act_eval(conv, act):
    """A copy of ``conv`` whose fused epilogue applies ``act`` (ReLU / LeakyReLU)."""
    assert not conv.training, "Fusion only for eval!"
    fused = copy.deepcopy(conv)
    if isinstance(act, torch.nn.ReLU):
        fused.act_type = Activation.ReLU
    elif isinstance(act, torch.nn.LeakyReLU):
        fused.act_type = Activation.LeakyReLU
        fused.act_alpha = act.negative_slope
    else:
        raise NotImplementedError
    return fused


def fold_sequential_eval(seq):
    """A copy of a ``SparseSequential`` (eval mode) in which every ``[sparse conv, BatchNorm1d, (ReLU |
    LeakyReLU)]`` run is ONE convolution with the normalisation folded into weight / bias and the
    activation in the kernel's epilogue -- the deployment form the two functions above exist for (reference
    ``quantization/utils.py:5-53`` + the fused ``intrinsic`` containers); nested SparseSequentials are folded
    recursively, everything else is kept."""
    from spconv_amd.pytorch.conv import SparseConvolution
    from spconv_amd.pytorch.modules import SparseSequential
    assert not seq.training, "Fusion only for eval!"
    mods = list(seq.children())
    out, i = [], 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, SparseSequential):
            m = fold_sequential_eval(m)
        elif (isinstance(m, SparseConvolution) and not m.conv1x1 and m.act_type == Activation.None_
              and i + 1 < len(mods) and type(mods[i + 1]) is torch.nn.BatchNorm1d
              and mods[i + 1].running_mean is not None):
            m = fuse_spconv_bn_eval(m, mods[i + 1])
            i += 1
            if i + 1 < len(mods) and isinstance(mods[i + 1], (torch.nn.ReLU, torch.nn.LeakyReLU)):
                m = fuse_spconv_act_eval(m, mods[i + 1])
                i += 1
        out.append(m)
        i += 1
    return SparseSequential(*out).eval()
