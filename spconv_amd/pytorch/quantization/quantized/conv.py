"""Statically quantised sparse convolution (reference:
``spconv/pytorch/quantization/quantized/conv.py:45-378``).

Input: a ``SparseConvTensor`` whose features are a per-tensor ``torch.qint8`` tensor
(zero point 0).  Weight: per-channel ``torch.qint8`` KRSC tensor (axis 0, zero points 0).
Output features: ``torch.qint8`` with scale ``self.scale``.  The convolution, the per-channel
rescale, bias, the optional residual input (``add_input``) and the activation run in one HIP
kernel (``spx_igemm_fwd_int8``), numerics as the reference's numpy formula
(``test/test_all_algo.py:272-287``)."""
from __future__ import annotations

from typing import List, Optional, Tuple, Union

import torch

from spconv_amd.pytorch.conv import SparseConvolution
from spconv_amd.pytorch.core import ConvAlgo, SparseConvTensor
from spconv_amd.pytorch.ops import Activation


class SparseConv(SparseConvolution):
    def __init__(self, ndim: int, in_channels: int, out_channels: int,
                 kernel_size: Union[int, List[int], Tuple[int, ...]] = 3,
                 stride: Union[int, List[int], Tuple[int, ...]] = 1,
                 padding: Union[int, List[int], Tuple[int, ...]] = 0,
                 dilation: Union[int, List[int], Tuple[int, ...]] = 1, groups: int = 1,
                 bias: bool = True, subm: bool = False,
                 output_padding: Union[int, List[int], Tuple[int, ...]] = 0,
                 transposed: bool = False, inverse: bool = False,
                 indice_key: Optional[str] = None, algo: Optional[ConvAlgo] = None,
                 fp32_accum: Optional[bool] = None, record_voxel_count: bool = False,
                 act_type=Activation.None_, act_alpha: float = 0, act_beta: float = 0,
                 device=None, dtype=None):
        super().__init__(ndim, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, bias=False, subm=subm, output_padding=output_padding,
                         transposed=transposed, inverse=inverse, indice_key=indice_key, algo=algo,
                         fp32_accum=fp32_accum, record_voxel_count=record_voxel_count,
                         act_type=act_type, act_alpha=act_alpha, act_beta=act_beta, device=device)
        self.scale = 1.0
        self.zero_point = 0
        self.eval()

    def _init_parameters(self, bias: bool, factory_kwargs) -> None:
        # no float Parameters: a per-channel qint8 weight and a float bias, set later through
        # set_weight_bias (reference quantized/conv.py:84-97)
        device = factory_kwargs.get("device")
        qweight = torch._empty_per_channel_affine_quantized(
            self.weight_shape, scales=torch.ones(self.out_channels),
            zero_points=torch.zeros(self.out_channels, dtype=torch.long), axis=0,
            dtype=torch.qint8, device=device)
        self.set_weight_bias(qweight, torch.zeros(self.out_channels, dtype=torch.float, device=device))

    def _get_name(self):
        return "QuantizedSparseConvolution"

    def set_weight_bias(self, w: torch.Tensor, b: Optional[torch.Tensor]) -> None:
        self._weight = w
        if b is None:   # the kernel always takes a bias (reference quantized/conv.py:354-359)
            self._bias = torch.zeros((w.shape[0],), dtype=torch.float32, device=w.device)
        else:
            self._bias = b

    def weight(self):
        return self._weight

    def bias(self):
        return self._bias

    def _apply(self, fn, *args, **kwargs):
        # the quantised weight is not a Parameter: move it with the module (.cuda() / .to())
        super()._apply(fn, *args, **kwargs)
        try:
            self._weight = fn(self._weight)
            self._bias = fn(self._bias)
        except Exception:   # dtype casts do not apply to quantised tensors
            pass
        return self

    @classmethod
    def from_float_conv(cls, conv: SparseConvolution, output_scale: float) -> "SparseConv":
        """Symmetric per-output-channel int8 quantisation of a float module's weight."""
        q = cls(conv.ndim, conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride,
                conv.padding, conv.dilation, conv.groups, subm=conv.subm,
                output_padding=conv.output_padding, transposed=conv.transposed,
                indice_key=conv.indice_key, algo=conv.algo, act_type=conv.act_type,
                act_alpha=conv.act_alpha, act_beta=conv.act_beta, device=conv.weight.device)
        w = conv.weight.detach().float()
        scales = w.abs().reshape(w.shape[0], -1).amax(dim=1).clamp_min(1e-8) / 127.0
        qw = torch.quantize_per_channel(w, scales, torch.zeros_like(scales, dtype=torch.long), 0,
                                        torch.qint8)
        q.set_weight_bias(qw, None if conv.bias is None else conv.bias.detach().float())
        q.scale = float(output_scale)
        return q

    def forward(self, input: SparseConvTensor, add_input: Optional[SparseConvTensor] = None):
        # reference quantized/conv.py:368-378
        inp_scale = input.q_scale()
        w_scales = self.weight().q_per_channel_scales().to(torch.float32)
        out_scale = self.scale
        channel_scale = (inp_scale * w_scales) / out_scale
        bias = self.bias() / out_scale
        return self._conv_forward(False, input, self.weight(), bias, add_input,
                                  channel_scale=channel_scale, output_scale=out_scale,
                                  name=self.name, sparse_unique_name=self._sparse_unique_name,
                                  act_type=self.act_type, act_alpha=self.act_alpha,
                                  act_beta=self.act_beta)
