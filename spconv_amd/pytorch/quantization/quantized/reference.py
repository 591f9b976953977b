"""Reference quantised sparse convolution: float kernel on a quantise -> dequantise'd weight
(reference ``quantization/quantized/reference.py:15-157``).  It is what ``convert_fx`` produces
first; a backend then lowers ``dequant -> SpConv -> quant`` to the int8 module."""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch
from torch.ao.nn.quantized.reference.modules.utils import ReferenceQuantizedModule

from spconv_amd.pytorch.conv import SparseConvolution, conv_ctor_kwargs
from spconv_amd.pytorch.core import SparseConvTensor


class SpConv(SparseConvolution, ReferenceQuantizedModule):
    __annotations__ = {"bias": Optional[torch.Tensor]}
    _IS_REFERENCE = True

    def __init__(self, *args, device=None, dtype=None, weight_qparams: Optional[Dict[str, Any]] = None,
                 **kwargs):
        SparseConvolution.__init__(self, *args, device=device, dtype=dtype, **kwargs)
        self._init_weight_qparams(weight_qparams, device)

    def forward(self, x: SparseConvTensor, add_input: Optional[SparseConvTensor] = None) -> SparseConvTensor:
        return self._conv_forward(self.training, x, self.get_weight(), self.bias, add_input=add_input)

    def _get_name(self):
        return "QuantizedSparseConv(Reference)"

    @classmethod
    def from_float(cls, float_conv, weight_qparams):
        ref = cls(**conv_ctor_kwargs(float_conv), device=float_conv.weight.device,
                  dtype=float_conv.weight.dtype, weight_qparams=weight_qparams)
        ref.weight = torch.nn.Parameter(float_conv.weight.detach())
        if float_conv.bias is not None:
            ref.bias = torch.nn.Parameter(float_conv.bias.detach())
        return ref
