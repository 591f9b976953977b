"""Module-swap tables of the eager-mode flow (reference ``quantization/qmapping.py:10-50``):
float / fused float -> QAT, and float / QAT -> statically quantised."""
from __future__ import annotations

from typing import Any, Callable, Dict

from spconv_amd.pytorch.conv import DEFAULT_SPARSE_CONV_TYPES, SparseConvolution
from spconv_amd.pytorch.quantization import intrinsic as snni
from spconv_amd.pytorch.quantization import quantized as snnq
from spconv_amd.pytorch.quantization.intrinsic import qat as snniqat
from spconv_amd.pytorch.quantization.intrinsic import quantized as snniq

STATIC_SPCONV_QUANT_MODULE_MAPPINGS: Dict[Callable, Any] = {t: snnq.SparseConv for t in DEFAULT_SPARSE_CONV_TYPES}
STATIC_SPCONV_QUANT_MODULE_MAPPINGS.update({
    SparseConvolution: snnq.SparseConv,
    snni.SpconvReLUNd: snniq.SparseConvReLU,
    snni.SpconvAddReLUNd: snniq.SparseConvAddReLU,
    snniqat.SparseConv: snnq.SparseConv,
    snniqat.SparseConvBn: snnq.SparseConv,
    snniqat.SparseConvBnReLU: snniq.SparseConvReLU,
    snniqat.SparseConvReLU: snniq.SparseConvReLU,
    snniqat.SparseConvBnAddReLU: snniq.SparseConvAddReLU,
    snniqat.SparseConvAddReLU: snniq.SparseConvAddReLU,
})

SPCONV_QAT_MODULE_MAPPINGS: Dict[Callable, Any] = {t: snniqat.SparseConv for t in DEFAULT_SPARSE_CONV_TYPES}
SPCONV_QAT_MODULE_MAPPINGS.update({
    SparseConvolution: snniqat.SparseConv,
    snni.SpconvReLUNd: snniqat.SparseConvReLU,
    snni.SpconvAddReLUNd: snniqat.SparseConvAddReLU,
    snni.SpconvBnNd: snniqat.SparseConvBn,
    snni.SpconvBnReLUNd: snniqat.SparseConvBnReLU,
    snni.SpconvBnAddReLUNd: snniqat.SparseConvBnAddReLU,
})


def get_spconv_qat_to_static_mapping():
    return STATIC_SPCONV_QUANT_MODULE_MAPPINGS


def get_spconv_fmod_to_qat_mapping():
    return SPCONV_QAT_MODULE_MAPPINGS
