"""torch.ao fx-graph-mode backend description for sparse convolutions (reference
``quantization/backend_cfg.py:39-649``, new-style pattern format of torch >= 2.0).

    backend_cfg = get_spconv_backend_config()
    prepare_cfg = get_spconv_prepare_custom_config()
    prepared = prepare_fx(model, get_default_spconv_qconfig_mapping(False), (),
                          backend_config=backend_cfg, prepare_custom_config=prepare_cfg)
    ... calibrate ...
    prepare_spconv_torch_inference(False)       # lowering tables: reference module -> int8 module
    converted = convert_fx(prepared, backend_config=backend_cfg)
    converted = remove_conv_add_dq(transform_qdq(converted))

Patterns: conv, conv+relu, conv+bn, conv+bn+relu, relu(conv(+bn) + residual), with their fused
float containers (``intrinsic``), QAT forms (``intrinsic.qat``) and the reference quantised module;
ToDense / SparseIdentity / pooling layers share the observer of their input."""
from __future__ import annotations

import operator
from typing import Dict, List, Optional, Tuple, Type

import torch
import torch.nn.functional as F
from torch import nn
from torch.ao.quantization.backend_config import (BackendConfig, BackendPatternConfig, DTypeConfig,
                                                  ObservationType, get_tensorrt_backend_config)
from torch.ao.quantization.fx.custom_config import ConvertCustomConfig, PrepareCustomConfig
from torch.ao.quantization.fx.match_utils import MatchAllNode

import spconv_amd.pytorch.conv as sconvmod
from spconv_amd.pytorch import pool as _pool
from spconv_amd.pytorch.modules import SparseBatchNorm, SparseIdentity, SparseReLU, SparseSyncBatchNorm
from spconv_amd.pytorch.quantization import intrinsic as snni
from spconv_amd.pytorch.quantization import quantized as snnq
from spconv_amd.pytorch.quantization.fuse_mapping import (fuse_conv_bn, fuse_conv_bn_add_relu,
                                                          fuse_conv_bn_relu)
from spconv_amd.pytorch.quantization.intrinsic import qat as snniqat
from spconv_amd.pytorch.quantization.intrinsic import quantized as snniq
from spconv_amd.pytorch.quantization.quantized import reference as snnqr

_OBS = ObservationType.OUTPUT_USE_DIFFERENT_OBSERVER_AS_INPUT
_SHARE = ObservationType.OUTPUT_SHARE_OBSERVER_WITH_INPUT

weighted_op_qint8_dtype_config = DTypeConfig(input_dtype=torch.qint8, output_dtype=torch.qint8,
                                             weight_dtype=torch.qint8, bias_dtype=torch.float)
non_weighted_op_qint8_dtype_config = DTypeConfig(input_dtype=torch.qint8, output_dtype=torch.qint8)
conv_dtype_configs = [weighted_op_qint8_dtype_config]

_ROOTS = [*sorted(sconvmod.DEFAULT_SPARSE_CONV_TYPES, key=lambda c: c.__name__), sconvmod.SparseConvolution]
_RELUS = [nn.ReLU, F.relu, SparseReLU]
_ADDS = [torch.add, operator.add]


def _to_dense_cls():
    from spconv_amd.pytorch import ToDense
    return ToDense


def _pool_layers():
    return [getattr(_pool, n) for n in dir(_pool)
            if n.startswith("Sparse") and isinstance(getattr(_pool, n), type)
            and issubclass(getattr(_pool, n), nn.Module)]


# ---- residual patterns, complex format: (relu, (add, (bn, conv), extra)) / (relu, (add, conv, extra))
def _bn_res_root(pattern):
    _, (_, (_, conv), _) = pattern
    return conv


def _bn_res_extra(pattern):
    _, (_, _, extra) = pattern
    return [extra]


def _res_root(pattern):
    _, (_, conv, _) = pattern
    return conv


def _res_extra(pattern):
    _, (_, _, extra) = pattern
    return [extra]


def _pair(container):
    def fuser(is_qat, m1, m2):
        return container(m1, m2)
    return fuser


def _add_relu_fuser(is_qat, relu, add_pattern):
    _, conv, _ = add_pattern
    return snni.SpconvAddReLUNd(conv, relu)


def _quantizable(pattern, root, qat=None):
    c = BackendPatternConfig(pattern).set_observation_type(_OBS).set_dtype_configs(conv_dtype_configs) \
        .set_root_module(root).set_reference_quantized_module(snnqr.SpConv)
    return c.set_qat_module(qat) if qat is not None else c


def _get_bn_spconv_configs(bn_cls, dtype_configs) -> List[BackendPatternConfig]:
    """conv + bn (+ relu) (+ residual) fusion patterns for one BatchNorm class."""
    out = []
    for root in _ROOTS:
        out.append(BackendPatternConfig((root, bn_cls)).set_dtype_configs(dtype_configs)
                   .set_fuser_method(fuse_conv_bn).set_fused_module(snni.SpconvBnNd))
        for relu in _RELUS:
            out.append(BackendPatternConfig((root, bn_cls, relu)).set_dtype_configs(dtype_configs)
                       .set_fuser_method(fuse_conv_bn_relu).set_fused_module(snni.SpconvBnReLUNd))
        for add in _ADDS:
            out.append(BackendPatternConfig()
                       ._set_pattern_complex_format((SparseReLU, (add, (bn_cls, root), MatchAllNode)))
                       .set_dtype_configs(dtype_configs).set_fuser_method(fuse_conv_bn_add_relu)
                       ._set_root_node_getter(_bn_res_root)._set_extra_inputs_getter(_bn_res_extra)
                       .set_fused_module(snni.SpconvBnAddReLUNd))
    return out


def _get_spconv_configs(dtype_configs) -> List[BackendPatternConfig]:
    out = []
    for bn in (SparseBatchNorm, nn.BatchNorm1d, SparseSyncBatchNorm):
        out += _get_bn_spconv_configs(bn, dtype_configs)
    fused_to_qat = [(snni.SpconvReLUNd, snniqat.SparseConvReLU), (snni.SpconvBnNd, snniqat.SparseConvBn),
                    (snni.SpconvBnReLUNd, snniqat.SparseConvBnReLU),
                    (snni.SpconvAddReLUNd, snniqat.SparseConvAddReLU),
                    (snni.SpconvBnAddReLUNd, snniqat.SparseConvBnAddReLU)]
    for root in _ROOTS:
        out.append(_quantizable(root, root, snniqat.SparseConv))                 # plain conv
        for relu in _RELUS:                                                      # conv + relu
            out.append(BackendPatternConfig((root, relu)).set_dtype_configs(dtype_configs)
                       .set_fuser_method(_pair(snni.SpconvReLUNd)).set_fused_module(snni.SpconvReLUNd))
        for add in _ADDS:                                                        # relu(conv + residual)
            out.append(BackendPatternConfig()
                       ._set_pattern_complex_format((SparseReLU, (add, root, MatchAllNode)))
                       .set_dtype_configs(dtype_configs).set_fuser_method(_add_relu_fuser)
                       ._set_root_node_getter(_res_root)._set_extra_inputs_getter(_res_extra)
                       .set_fused_module(snni.SpconvAddReLUNd))
    root = sconvmod.SparseConvolution
    out.append(_quantizable(snniqat.SparseConv, root))
    for fused, qat in fused_to_qat:
        if fused in (snni.SpconvReLUNd, snni.SpconvAddReLUNd):
            out.append(_quantizable(fused, root, qat))          # bn-free containers quantise directly
        else:
            out.append(BackendPatternConfig(fused).set_dtype_configs(dtype_configs).set_qat_module(qat))
        out.append(_quantizable(qat, root))
    return out


def _get_share_observer_ops(dtype_configs) -> List[BackendPatternConfig]:
    return [BackendPatternConfig(m).set_observation_type(_SHARE).set_dtype_configs(dtype_configs)
            for m in (_to_dense_cls(), SparseIdentity, *_pool_layers())]


SPCONV_STATIC_LOWER_FUSED_MODULE_MAP: Dict[Type[nn.Module], Tuple[Type[nn.Module], Type[nn.Module]]] = {
    snni.SpconvReLUNd: (snnqr.SpConv, snniq.SparseConvReLU),
    snni.SpconvAddReLUNd: (snnqr.SpConv, snniq.SparseConvAddReLU),
}
SPCONV_STATIC_LOWER_MODULE_MAP: Dict[Type[nn.Module], Type[nn.Module]] = {snnqr.SpConv: snnq.SparseConv}


def get_spconv_backend_config(additional_bns: Optional[List[Type[nn.Module]]] = None) -> BackendConfig:
    """The tensorrt-style (symmetric int8) torch.ao backend config + the sparse patterns."""
    cfg = get_tensorrt_backend_config().set_backend_pattern_configs(
        _get_spconv_configs(conv_dtype_configs) + _get_share_observer_ops([non_weighted_op_qint8_dtype_config]))
    for bn in additional_bns or []:
        cfg.set_backend_pattern_configs(_get_bn_spconv_configs(bn, conv_dtype_configs))
    return cfg


def get_spconv_prepare_custom_config(additional_bns: Optional[List[Type[nn.Module]]] = None) -> PrepareCustomConfig:
    """Sparse layers are leaves of the fx trace (their forward branches on tensor contents)."""
    cfg = PrepareCustomConfig()
    cfg.non_traceable_module_classes = [*_ROOTS, SparseReLU, SparseBatchNorm, SparseSyncBatchNorm,
                                        *(additional_bns or [])]
    return cfg


def get_spconv_convert_custom_config() -> ConvertCustomConfig:
    cfg = ConvertCustomConfig()
    cfg.set_observed_to_quantized_mapping(snni.SpconvReLUNd, snniq.SparseConvReLU)
    cfg.set_observed_to_quantized_mapping(snni.SpconvAddReLUNd, snniq.SparseConvReLU)
    return cfg


def prepare_spconv_torch_inference(with_linear: bool = False) -> None:
    """Registers the sparse modules in torch.ao's reference -> native lowering tables so that
    ``convert_fx`` ends in the int8 modules (reference backend_cfg.py:631-649).  ``with_linear``
    is accepted for signature parity; dense Linear layers keep torch's own lowering."""
    from torch.ao.quantization.fx import _lower_to_native_backend as low
    fused = dict(SPCONV_STATIC_LOWER_FUSED_MODULE_MAP)
    two_inputs = getattr(low, "STATIC_LOWER_FUSED_MODULE_TWO_INPUTS_MAP", None)
    if two_inputs is not None:
        # torch >= 2.1 lowers (dequantize, dequantize) -> fused module -> quantize in its own pass;
        # the single-input pass insists on exactly one argument
        two_inputs[snni.SpconvAddReLUNd] = fused.pop(snni.SpconvAddReLUNd)
    low.STATIC_LOWER_FUSED_MODULE_MAP.update(fused)
    low.STATIC_LOWER_MODULE_MAP.update(SPCONV_STATIC_LOWER_MODULE_MAP)
