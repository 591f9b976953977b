"""Quantised conv + ReLU and conv + residual add + ReLU: one int8 kernel launch each, the
activation and the residual run in the kernel's epilogue (reference
``quantization/intrinsic/quantized/conv_relu.py:17-107``)."""
from __future__ import annotations

from typing import Optional

import torch

from spconv_amd.pytorch.core import SparseConvTensor
from spconv_amd.pytorch.ops import Activation
from spconv_amd.pytorch.quantization import intrinsic as snni
from spconv_amd.pytorch.quantization.quantized.conv import SparseConv

__all__ = ["SparseConvReLU", "SparseConvAddReLU"]


class SparseConvReLU(SparseConv):
    _FLOAT_MODULE = snni.SpconvReLUNd

    def forward(self, input: SparseConvTensor, add_input: Optional[SparseConvTensor] = None):
        out_scale = self.scale
        channel_scale = (input.q_scale() * self.weight().q_per_channel_scales().to(torch.float32)) / out_scale
        return self._conv_forward(False, input, self.weight(), self.bias() / out_scale, add_input,
                                  channel_scale=channel_scale, output_scale=out_scale,
                                  act_type=Activation.ReLU)

    def _get_name(self):
        return "QuantizedSparseConvReLU"

    @classmethod
    def from_reference(cls, ref_qconv, output_scale, output_zero_point):
        assert type(ref_qconv) != snni.SpconvBnReLUNd, \
            "BatchNorm1d should be fused into the conv before converting to reference module"
        return super().from_reference(ref_qconv[0], output_scale, output_zero_point)


class SparseConvAddReLU(SparseConvReLU):
    _FLOAT_MODULE = snni.SpconvAddReLUNd

    def _get_name(self):
        return "QuantizedSparseConvAddReLU"
