"""Quantization-aware-training forms of the sparse convolution (reference
``quantization/intrinsic/qat/modules.py:20-713``, themselves the sparse counterparts of
``torch.ao.nn.qat`` / ``torch.ao.nn.intrinsic.qat`` conv modules).

* ``SparseConv`` (+ ``SparseConvReLU``, ``SparseConvAddReLU``): weight goes through a
  fake-quantizer before the convolution.
* ``SparseConvBn`` (+ ``...ReLU``, ``...AddReLU``): conv and BatchNorm trained as one op.  The
  weight is scaled by ``gamma / running_std`` BEFORE fake-quantization (so that the quantizer
  sees what inference will use), the output is un-scaled and handed to the real BatchNorm, which
  keeps updating its statistics (single forward pass; arXiv 1806.08342 section 3.2).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.ao.nn.intrinsic as nni
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import init
from torch.nn.parameter import Parameter

from spconv_amd.pytorch.conv import SparseConvolution, conv_ctor_kwargs
from spconv_amd.pytorch.core import SparseConvTensor
from spconv_amd.pytorch.quantization import intrinsic as snni
from spconv_amd.pytorch.quantization.utils import fuse_spconv_bn_weights


def _float_conv_of(mod: SparseConvolution) -> SparseConvolution:
    conv = SparseConvolution(**conv_ctor_kwargs(mod))
    conv.weight = Parameter(mod.weight.detach())
    if mod.bias is not None:
        conv.bias = Parameter(mod.bias.detach())
    return conv


class SparseConv(SparseConvolution):
    """Sparse convolution with a FakeQuantize module on its weight."""
    _FLOAT_MODULE = SparseConvolution
    _FLOAT_CONV_MODULE = SparseConvolution
    _FLOAT_RELU_MODULE = None

    def __init__(self, *args, qconfig=None, device=None, dtype=None, **kwargs):
        super().__init__(*args, device=device, dtype=dtype, **kwargs)
        assert qconfig, "qconfig must be provided for QAT module"
        self.qconfig = qconfig
        self.weight_fake_quant = qconfig.weight(factory_kwargs={"device": device, "dtype": dtype})

    def forward(self, input):
        return self._conv_forward(self.training, input, self.weight_fake_quant(self.weight), self.bias)

    @classmethod
    def from_float(cls, mod):
        assert isinstance(mod, cls._FLOAT_MODULE), \
            f"qat.{cls.__name__}.from_float only works for {cls._FLOAT_MODULE.__name__} not {type(mod).__qualname__}"
        assert getattr(mod, "qconfig", None), "Input float module must have a valid qconfig"
        qconfig = mod.qconfig
        if isinstance(mod, nni._FusedModule):
            mod = mod[0]
        qat = cls(**conv_ctor_kwargs(mod), qconfig=qconfig)
        qat.weight = mod.weight
        qat.bias = mod.bias
        return qat

    def to_float(self):
        conv = _float_conv_of(self)
        cls = type(self)
        if cls._FLOAT_RELU_MODULE is None:
            return conv
        fused = cls._FLOAT_MODULE(conv, cls._FLOAT_RELU_MODULE())
        fused.train(self.training)
        return fused


class SparseConvReLU(SparseConv, nni._FusedModule):
    _FLOAT_MODULE = snni.SpconvReLUNd
    _FLOAT_RELU_MODULE = nn.ReLU

    def forward(self, input):
        x = self._conv_forward(self.training, input, self.weight_fake_quant(self.weight), self.bias)
        return x.replace_feature(F.relu(x.features))


class SparseConvAddReLU(SparseConv, nni._FusedModule):
    _FLOAT_MODULE = snni.SpconvAddReLUNd
    _FLOAT_RELU_MODULE = nn.ReLU

    def forward(self, input, add_input):
        x = self._conv_forward(self.training, input, self.weight_fake_quant(self.weight), self.bias,
                               add_input=add_input)
        return x.replace_feature(F.relu(x.features))


class _SparseConvBn(SparseConvolution, nni._FusedModule):
    _version = 2
    _FLOAT_MODULE = snni.SpconvBnNd
    _FLOAT_CONV_MODULE = SparseConvolution
    _FLOAT_BN_MODULE = nn.BatchNorm1d
    _FLOAT_RELU_MODULE = None
    _FUSED_FLOAT_MODULE = snni.SpconvReLUNd      # float form once bn is folded (with relu)

    def __init__(self, *args, bias: bool = True, eps=1e-05, momentum=0.1, freeze_bn=False, qconfig=None,
                 **kwargs):
        # the conv itself is bias-free: the bias is added after un-scaling, before the BatchNorm
        SparseConvolution.__init__(self, *args, bias=False, **kwargs)
        assert qconfig, "qconfig must be provided for QAT module"
        self.qconfig = qconfig
        self.freeze_bn = freeze_bn if self.training else True
        self.bn = nn.BatchNorm1d(self.out_channels, eps, momentum, True, True)
        self.weight_fake_quant = self.qconfig.weight()
        if bias:
            self.bias = Parameter(torch.empty(self.out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_bn_parameters()
        if self.training and not freeze_bn:
            self.update_bn_stats()
        else:
            self.freeze_bn_stats()

    def reset_running_stats(self):
        self.bn.reset_running_stats()

    def reset_bn_parameters(self):
        self.bn.reset_running_stats()
        init.uniform_(self.bn.weight)
        init.zeros_(self.bn.bias)
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            init.uniform_(self.bias, -1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))

    def update_bn_stats(self):
        self.freeze_bn = False
        self.bn.training = True
        return self

    def freeze_bn_stats(self):
        self.freeze_bn = True
        self.bn.training = False
        return self

    def _forward(self, input: SparseConvTensor, add_input: Optional[SparseConvTensor] = None):
        assert self.bn.running_var is not None
        running_std = torch.sqrt(self.bn.running_var + self.bn.eps)
        scale = self.bn.weight / running_std                              # per output channel
        w_shape = [-1] + [1] * (self.weight.ndim - 1)
        scaled_w = self.weight_fake_quant(self.weight * scale.reshape(w_shape))
        zero_bias = torch.zeros(self.out_channels, device=scaled_w.device, dtype=input.features.dtype)
        y = self._conv_forward(self.training, input, scaled_w, zero_bias)
        feat = y.features / scale                                          # back to the un-folded conv
        if self.bias is not None:
            feat = feat + self.bias
        feat = self.bn(feat)
        if add_input is not None:
            feat = feat + add_input.features
        return y.replace_feature(feat)

    def forward(self, input):
        return self._forward(input)

    def train(self, mode=True):
        """A frozen BatchNorm keeps its eval behaviour under ``model.train()``."""
        self.training = mode
        if not self.freeze_bn:
            for m in self.children():
                m.train(mode)
        return self

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        # version 1 kept the BatchNorm tensors on the module itself (gamma, beta, running_*)
        if local_metadata.get("version", None) in (None, 1):
            for new, old in (("bn.weight", "gamma"), ("bn.bias", "beta"),
                             ("bn.running_mean", "running_mean"), ("bn.running_var", "running_var"),
                             ("bn.num_batches_tracked", "num_batches_tracked")):
                if prefix + old in state_dict:
                    state_dict[prefix + new] = state_dict.pop(prefix + old)
                elif prefix + new not in state_dict and strict:
                    missing_keys.append(prefix + new)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys,
                                      unexpected_keys, error_msgs)

    @classmethod
    def from_float(cls, mod):
        assert type(mod) == cls._FLOAT_MODULE, \
            f"qat.{cls.__name__}.from_float only works for {cls._FLOAT_MODULE.__name__}"
        assert getattr(mod, "qconfig", None), "Input float module must have a valid qconfig"
        conv, bn = mod[0], mod[1]
        kw = conv_ctor_kwargs(conv)
        qat = cls(**kw, eps=bn.eps, momentum=bn.momentum, freeze_bn=False, qconfig=mod.qconfig)
        qat.weight = conv.weight
        qat.bias = conv.bias
        qat.bn.weight, qat.bn.bias = bn.weight, bn.bias
        qat.bn.running_mean, qat.bn.running_var = bn.running_mean, bn.running_var
        qat.bn.num_batches_tracked = bn.num_batches_tracked
        return qat

    def to_float(self):
        cls = type(self)
        conv = _float_conv_of(self)
        conv.weight, conv.bias = fuse_spconv_bn_weights(conv.weight, conv.bias, self.bn.running_mean,
                                                        self.bn.running_var, self.bn.eps,
                                                        self.bn.weight, self.bn.bias)
        if cls._FLOAT_RELU_MODULE is None:
            conv.train(self.training)
            return conv
        fused = cls._FUSED_FLOAT_MODULE(conv, cls._FLOAT_RELU_MODULE())
        fused.train(self.training)
        return fused


class SparseConvBn(_SparseConvBn):
    _FLOAT_MODULE = snni.SpconvBnNd


class SparseConvBnReLU(_SparseConvBn):
    _FLOAT_MODULE = snni.SpconvBnReLUNd
    _FLOAT_RELU_MODULE = nn.ReLU

    def forward(self, input):
        x = self._forward(input)
        return x.replace_feature(F.relu(x.features))


class SparseConvBnAddReLU(_SparseConvBn):
    _FLOAT_MODULE = snni.SpconvBnAddReLUNd
    _FLOAT_RELU_MODULE = nn.ReLU
    _FUSED_FLOAT_MODULE = snni.SpconvAddReLUNd

    def forward(self, input, add_input):
        x = self._forward(input, add_input)
        return x.replace_feature(F.relu(x.features))
