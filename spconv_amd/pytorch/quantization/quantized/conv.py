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

from torch.ao.nn.quantized.modules.utils import _quantize_weight

from spconv_amd.pytorch.conv import SparseConvolution, conv_ctor_kwargs
from spconv_amd.pytorch.core import ConvAlgo, SparseConvTensor
from spconv_amd.pytorch.ops import Activation
from spconv_amd.pytorch.quantization.utils import fuse_spconv_bn_weights


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
                 name=None, device=None, dtype=None):
        super().__init__(ndim, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, bias=False, subm=subm, output_padding=output_padding,
                         transposed=transposed, inverse=inverse, indice_key=indice_key, algo=algo,
                         fp32_accum=fp32_accum, record_voxel_count=record_voxel_count,
                         act_type=act_type, act_alpha=act_alpha, act_beta=act_beta, name=name,
                         device=device)
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

    # ---- serialisation (reference quantized/conv.py:120-148): weight / bias / scale / zero_point
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        destination[prefix + "weight"] = self._weight
        destination[prefix + "bias"] = self._bias
        destination[prefix + "scale"] = torch.tensor(self.scale)
        destination[prefix + "zero_point"] = torch.tensor(self.zero_point)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        self.set_weight_bias(state_dict.pop(prefix + "weight"), state_dict.pop(prefix + "bias"))
        self.scale = float(state_dict.pop(prefix + "scale"))
        self.zero_point = int(state_dict.pop(prefix + "zero_point"))
        super()._load_from_state_dict(state_dict, prefix, local_metadata, False, missing_keys,
                                      unexpected_keys, error_msgs)

    # ---- conversion from float / QAT / reference modules (reference quantized/conv.py:153-258)
    _FLOAT_MODULE = SparseConvolution

    @classmethod
    def get_qconv(cls, mod, activation_post_process, weight_post_process=None):
        """Quantised module from a float conv ``mod`` and its observers."""
        if weight_post_process is None:
            weight_post_process = mod.qconfig.weight()
        weight_post_process = weight_post_process.to(mod.weight.device)
        weight_post_process(mod.weight)
        assert weight_post_process.dtype == torch.qint8, "Weight observer must have a dtype of qint8"
        qweight = _quantize_weight(mod.weight.float(), weight_post_process)
        kw = conv_ctor_kwargs(mod)
        kw.pop("bias")
        qconv = cls(**kw, device=mod.weight.device)
        qconv.set_weight_bias(qweight, None if mod.bias is None else mod.bias.detach().float())
        if activation_post_process is None or activation_post_process.dtype == torch.float:
            return qconv
        act_scale, act_zp = activation_post_process.calculate_qparams()
        qconv.scale = float(act_scale)
        qconv.zero_point = int(act_zp)
        return qconv

    @classmethod
    def from_float(cls, mod):
        """From an observed float module (PTQ), a fused float container, or a QAT module."""
        import torch.ao.nn.intrinsic as nni
        if hasattr(mod, "weight_fake_quant"):                         # QAT module
            if hasattr(mod, "bn"):                                    # conv + bn trained together
                mod.weight, mod.bias = fuse_spconv_bn_weights(
                    mod.weight, mod.bias, mod.bn.running_mean, mod.bn.running_var, mod.bn.eps,
                    mod.bn.weight, mod.bn.bias)
            assert hasattr(mod, "activation_post_process"), "Input QAT module must have observer attached"
            return cls.get_qconv(mod, mod.activation_post_process, mod.weight_fake_quant)
        assert isinstance(mod, (SparseConvolution, nni._FusedModule)),             f"nnq.{cls.__name__}.from_float only works for sparse convolutions but got: {type(mod)}"
        assert hasattr(mod, "qconfig"), "Input float module must have qconfig defined."
        act_pp = getattr(mod, "activation_post_process", None)
        qconfig = mod.qconfig
        if isinstance(mod, nni._FusedModule):
            mod = mod[0]
        return cls.get_qconv(mod, act_pp, qconfig.weight())

    @classmethod
    def from_reference(cls, ref_qconv, output_scale, output_zero_point):
        """From a reference quantised module (``quantized.reference.SpConv``)."""
        kw = conv_ctor_kwargs(ref_qconv)
        kw.pop("bias")
        qconv = cls(**kw, device=ref_qconv.weight.device)
        qconv.set_weight_bias(ref_qconv.get_quantized_weight(),
                              None if ref_qconv.bias is None else ref_qconv.bias.detach().float())
        qconv.scale = float(output_scale)
        qconv.zero_point = int(output_zero_point)
        return qconv

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
