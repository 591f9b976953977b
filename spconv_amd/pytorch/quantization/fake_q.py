"""Observers / fake-quantizers that see through SparseConvTensor, and the default symmetric
int8 qconfigs (reference ``quantization/fake_q.py:27-176``): per-tensor symmetric activations
(zero point 0, what the int8 kernel's epilogue assumes), per-channel weights."""
from __future__ import annotations

from typing import Any, Dict

import torch
from torch.ao.quantization.fake_quantize import (FakeQuantize, FixedQParamsFakeQuantize,
                                                 FusedMovingAvgObsFakeQuantize,
                                                 default_fused_per_channel_wt_fake_quant,
                                                 default_weight_fake_quant)
from torch.ao.quantization.observer import (HistogramObserver, MinMaxObserver,
                                            MovingAverageMinMaxObserver,
                                            default_per_channel_weight_observer,
                                            default_placeholder_observer, default_weight_observer)
from torch.ao.quantization.qconfig import QConfig, QConfigAny, default_reuse_input_qconfig
from torch.ao.quantization.qconfig_mapping import _FIXED_QPARAMS_OP_TO_OBSERVER, QConfigMapping

from spconv_amd.pytorch.core import SparseConvTensor

__all__ = ["get_default_spconv_trt_ptq_qconfig", "get_default_spconv_trt_qat_qconfig",
           "get_default_spconv_qconfig_mapping"]


def _sparse_aware(base):
    """Subclass of an observer / fake-quant whose forward also takes a SparseConvTensor: the
    features are observed (and fake-quantised), the tensor keeps its coordinates."""

    class _Sparse(base):
        def forward(self, input):
            if isinstance(input, SparseConvTensor):
                return input.replace_feature(super().forward(input.features))
            return super().forward(input)

    _Sparse.__name__ = _Sparse.__qualname__ = "Sparse" + base.__name__
    return _Sparse


SparseFusedMovingAvgObsFakeQuantize = _sparse_aware(FusedMovingAvgObsFakeQuantize)
SparseMovingAvgObsFakeQuantize = _sparse_aware(FakeQuantize)
SparseMovingAvgObsFakeQuantize.__name__ = SparseMovingAvgObsFakeQuantize.__qualname__ = \
    "SparseMovingAvgObsFakeQuantize"
SparseHistogramObserver = _sparse_aware(HistogramObserver)
SparseMinMaxObserver = _sparse_aware(MinMaxObserver)

_ACT = dict(quant_min=-128, quant_max=127, dtype=torch.qint8, reduce_range=False,
            qscheme=torch.per_tensor_symmetric, eps=2 ** -12)

default_symmetric_spconv_ptq_qconfig = QConfig(
    activation=SparseHistogramObserver.with_args(**_ACT),
    weight=default_per_channel_weight_observer)

default_symmetric_spconv_qat_qconfig = QConfig(
    activation=SparseFusedMovingAvgObsFakeQuantize.with_args(observer=MovingAverageMinMaxObserver, **_ACT),
    weight=default_fused_per_channel_wt_fake_quant)


def get_default_spconv_trt_ptq_qconfig(backend=None, version=0):
    return default_symmetric_spconv_ptq_qconfig


def get_default_spconv_trt_qat_qconfig(backend=None, version=0):
    return default_symmetric_spconv_qat_qconfig


def get_default_spconv_qconfig_mapping(is_qat: bool, backend: str = "fbgemm", version: int = 0) -> QConfigMapping:
    """The default QConfigMapping of torch.ao with the symmetric sparse-aware qconfig as the
    global one (reference fake_q.py:108-176)."""
    qconfig = default_symmetric_spconv_qat_qconfig if is_qat else default_symmetric_spconv_ptq_qconfig
    per_tensor_w = default_weight_fake_quant if is_qat else default_weight_observer
    # per-channel weight observers do not work with transposed convolutions on fbgemm / x86
    qconfig_transpose = QConfig(activation=qconfig.activation, weight=per_tensor_w) \
        if backend in ("fbgemm", "x86") else qconfig
    qconfig_layernorm = QConfig(activation=qconfig.activation, weight=default_placeholder_observer)
    mapping = QConfigMapping().set_global(qconfig).set_object_type("reshape", default_reuse_input_qconfig)
    for t in (torch.nn.ConvTranspose1d, torch.nn.ConvTranspose2d, torch.nn.ConvTranspose3d,
              torch.nn.functional.conv_transpose1d, torch.nn.functional.conv_transpose2d,
              torch.nn.functional.conv_transpose3d):
        mapping.set_object_type(t, qconfig_transpose)
    mapping.set_object_type(torch.nn.functional.layer_norm, qconfig_layernorm)
    mapping.set_object_type(torch.nn.LayerNorm, qconfig_layernorm)
    fixed: Dict[Any, QConfigAny] = {}
    for op, observer in _FIXED_QPARAMS_OP_TO_OBSERVER.items():
        if observer not in fixed:
            act = FixedQParamsFakeQuantize.with_args(observer=observer) if is_qat else observer
            fixed[observer] = QConfig(activation=act, weight=per_tensor_w)
        mapping.set_object_type(op, fixed[observer])
    if backend == "onednn":
        for t in (torch.nn.Linear, torch.nn.LeakyReLU, torch.nn.functional.leaky_relu, torch.nn.Tanh,
                  torch.nn.functional.tanh):
            mapping.set_object_type(t, qconfig)
    return mapping
