"""Sparse convolution modules.

API parity target: ``spconv/pytorch/conv.py`` of the reference --
``SparseConvolution`` (``conv.py:563-764``) and the 16 public classes
(``conv.py:767-1308``): same constructor signatures, KRSC ``weight``
``[K, *ksize, C]`` (``conv.py:136-139``), ``bias [K]``, kaiming-uniform init
(``conv.py:726-750``), rulebook sharing through ``indice_key`` with the same
validity checks (``conv.py:519-560``), 1x1 shortcut (``conv.py:225-241``),
training bias added outside the kernel, fused bias/activation at inference
(``conv.py:176-184,492-493``), checkpoint layout hook (``conv.py:648-683``).

One difference by design: both ``ConvAlgo`` values run the same
output-stationary implicit-GEMM kernels; ``algo`` only selects which
bookkeeping object (``IndiceData`` / ``ImplicitGemmIndiceData``) is stored.
"""
from __future__ import annotations

import math
import contextlib
import sys
import time
from typing import List, Optional, Tuple, Union

import numpy as np
import torch
from torch.nn import functional as F
from torch.nn import init
from torch.nn.init import calculate_gain
from torch.nn.parameter import Parameter

from spconv_amd import constants
from spconv_amd.constants import MODULE_DO_SORT, SAVED_WEIGHT_LAYOUT
from spconv_amd.tools import save_debug_data
from spconv_amd.pytorch import functional as Fsp
from spconv_amd.pytorch import ops
from spconv_amd.pytorch import prefetch
from spconv_amd.pytorch.core import (ConvAlgo, ImplicitGemmIndiceData, IndiceData, Rulebook,
                                     SparseConvTensor, expand_nd)
from spconv_amd.pytorch.modules import SparseModule
from spconv_amd.pytorch.ops import Activation

_MAX_NUM_VOXELS_DURING_TRAINING = "max_num_voxels_during_training"


def _apply_act(x: torch.Tensor, act_type, act_alpha: float, act_beta: float):
    if act_type == Activation.None_:
        return x
    if act_type == Activation.ReLU:
        return F.relu(x)
    if act_type == Activation.Sigmoid:
        return F.sigmoid(x)
    if act_type == Activation.LeakyReLU:
        return F.leaky_relu(x, act_alpha)
    raise NotImplementedError


class SparseConvolution(SparseModule):
    __constants__ = ["stride", "padding", "dilation", "groups", "bias", "subm", "inverse",
                     "transposed", "output_padding"]

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
                 large_kernel_fast_algo: bool = False, name=None, device=None, dtype=None):
        super().__init__(name=name)
        assert groups == 1, "don't support groups for now"
        self.ndim = ndim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = expand_nd(ndim, kernel_size)
        self.stride = expand_nd(ndim, stride)
        self.dilation = expand_nd(ndim, dilation)
        self.padding = expand_nd(ndim, padding)
        self.output_padding = expand_nd(ndim, output_padding)
        kv = int(np.prod(self.kernel_size))
        self.conv1x1 = kv == 1
        if not subm:
            self.conv1x1 &= int(np.prod(self.stride)) == 1
            if self.conv1x1:
                assert self.padding == [0] * ndim, "padding must be zero for 1x1 conv (k=1,s=1)"
        self.transposed = transposed
        self.inverse = inverse
        self.groups = groups
        self.subm = subm
        self.indice_key = indice_key
        self.record_voxel_count = record_voxel_count
        if algo is None:
            # reference default (conv.py:110-120): implicit GEMM when the kernel volume fits
            # the mask width, Native otherwise
            limit = 128 if large_kernel_fast_algo else 32
            algo = ConvAlgo.MaskImplicitGemm if kv <= limit else ConvAlgo.Native
        self.algo = algo
        self.fp32_accum = fp32_accum  # accumulation is always fp32 here
        self.weight_shape = [out_channels, *self.kernel_size, in_channels]  # KRSC
        self.act_type = act_type
        self.act_alpha = act_alpha
        self.act_beta = act_beta
        if self.conv1x1:
            assert act_type == Activation.None_, "conv1x1 don't support fused act"

        factory_kwargs = {"device": device, "dtype": dtype}
        if record_voxel_count and not self.subm and not self.inverse:
            self.register_buffer(_MAX_NUM_VOXELS_DURING_TRAINING,
                                 torch.zeros(1, dtype=torch.int32, device=device))
        self._init_parameters(bias, factory_kwargs)

    def _init_parameters(self, bias: bool, factory_kwargs) -> None:
        """Float weight / bias Parameters (the quantised module overrides this, like the
        reference's SparseConvolutionBase / SparseConvolution split, conv.py:63-147,563-764)."""
        self.weight = Parameter(torch.zeros(*self.weight_shape, **factory_kwargs))
        if bias:
            self.bias = Parameter(torch.zeros(self.out_channels, **factory_kwargs))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()
        self._register_load_state_dict_pre_hook(self._load_weight_different_layout)

    # ------------------------------------------------------------- parameters
    def get_max_num_voxels(self) -> Optional[torch.Tensor]:
        return getattr(self, _MAX_NUM_VOXELS_DURING_TRAINING, None)

    def is_inverseable(self):
        return self.indice_key is not None and not self.subm

    def _load_weight_different_layout(self, state_dict, prefix, local_metadata, strict,
                                      missing_keys, unexpected_keys, error_msgs):
        name = prefix + _MAX_NUM_VOXELS_DURING_TRAINING
        if (self.record_voxel_count and not self.subm and not self.inverse
                and name not in state_dict):
            state_dict[name] = torch.zeros(1, dtype=torch.int32)
        if not SAVED_WEIGHT_LAYOUT or SAVED_WEIGHT_LAYOUT == "KRSC":
            return
        key = prefix + "weight"
        assert key in state_dict
        nd = self.ndim
        if SAVED_WEIGHT_LAYOUT == "RSKC":    # [*ksize, K, C] -> KRSC
            state_dict[key] = state_dict[key].permute(nd, *range(nd), nd + 1).contiguous()
        elif SAVED_WEIGHT_LAYOUT == "RSCK":  # [*ksize, C, K] -> KRSC
            state_dict[key] = state_dict[key].permute(nd + 1, *range(nd), nd).contiguous()

    def extra_repr(self):
        s = "{in_channels}, {out_channels}, kernel_size={kernel_size}, stride={stride}"
        if self.padding != [0] * len(self.padding):
            s += ", padding={padding}"
        if self.dilation != [1] * len(self.dilation):
            s += ", dilation={dilation}"
        if self.output_padding != [0] * len(self.output_padding):
            s += ", output_padding={output_padding}"
        if self.groups != 1:
            s += ", groups={groups}"
        if self.bias is None:
            s += ", bias=False"
        if self.algo is not None:
            s += f", algo={self.algo}"
        if self.act_type != Activation.None_:
            s += f", act={self.act_type}"
        return s.format(**self.__dict__)

    def _calculate_fan_in_and_fan_out(self):
        receptive_field_size = int(np.prod(self.kernel_size))
        return self.in_channels * receptive_field_size, self.out_channels * receptive_field_size

    def _custom_kaiming_uniform_(self, tensor, a=0, mode="fan_in", nonlinearity="leaky_relu"):
        """torch.nn.init.kaiming_uniform_ for the KRSC layout (fan computed from channels)."""
        mode = mode.lower()
        if mode not in ("fan_in", "fan_out"):
            raise ValueError(f"Mode {mode} not supported, please use one of ['fan_in', 'fan_out']")
        fan_in, fan_out = self._calculate_fan_in_and_fan_out()
        fan = fan_in if mode == "fan_in" else fan_out
        bound = math.sqrt(3.0) * calculate_gain(nonlinearity, a) / math.sqrt(fan)
        with torch.no_grad():
            return tensor.uniform_(-bound, bound)

    def reset_parameters(self):
        self._custom_kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = self._calculate_fan_in_and_fan_out()
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    # ---------------------------------------------------------------- forward
    def forward(self, input: SparseConvTensor, add_input: Optional[SparseConvTensor] = None):
        return self._conv_forward(self.training, input, self.weight, self.bias, add_input,
                                  name=self.name, sparse_unique_name=self._sparse_unique_name,
                                  act_type=self.act_type, act_alpha=self.act_alpha,
                                  act_beta=self.act_beta)

    def _check_subm_reuse_valid(self, inp: SparseConvTensor, spatial_shape: List[int], datas):
        assert datas.is_subm, "only support reuse subm indices"
        if self.kernel_size != datas.ksize:
            raise ValueError(f"subm with same indice_key must have same kernel size, expect "
                             f"{datas.ksize}, this layer {self.kernel_size}")
        if self.dilation != datas.dilation:
            raise ValueError(f"subm with same indice_key must have same dilation, expect "
                             f"{datas.dilation}, this layer {self.dilation}")
        if inp.spatial_shape != datas.spatial_shape:
            raise ValueError(f"subm with same indice_key must have same spatial structure, "
                             f"expect {datas.spatial_shape}, input {spatial_shape}")
        if inp.indices.shape[0] != datas.indices.shape[0]:
            raise ValueError(f"subm with same indice_key must have same num of indices, expect "
                             f"{datas.indices.shape[0]}, input {inp.indices.shape[0]}")

    def _check_inverse_reuse_valid(self, inp: SparseConvTensor, spatial_shape: List[int], datas):
        if self.kernel_size != datas.ksize:
            raise ValueError(f"Inverse with same indice_key must have same kernel size, expect "
                             f"{datas.ksize}, this layer {self.kernel_size}, "
                             "please check Inverse Convolution in docs/USAGE.md.")
        if inp.spatial_shape != datas.out_spatial_shape:
            raise ValueError(f"Inverse with same indice_key must have same spatial structure "
                             f"(spatial shape), expect {datas.spatial_shape}, input "
                             f"{spatial_shape}, please check Inverse Convolution in docs/USAGE.md.")
        if inp.indices.shape[0] != datas.out_indices.shape[0]:
            raise ValueError(f"Inverse with same indice_key must have same num of indices, "
                             f"expect {datas.indices.shape[0]}, input {inp.indices.shape[0]}, "
                             "please check Inverse Convolution in .")

    def _make_indice_data(self, rb: Rulebook, indices, spatial_shape, out_spatial_shape, algo):
        common = dict(is_subm=self.subm, algo=algo, ksize=self.kernel_size, stride=self.stride,
                      dilation=self.dilation, padding=self.padding, rulebook=rb)
        if algo == ConvAlgo.Native:
            return IndiceData(rb.out_indices, indices, rb.pair_native, rb.num_per_loc,
                              spatial_shape, out_spatial_shape, **common)
        masks = [np.array([0xffffffff], dtype=np.uint32)]
        return ImplicitGemmIndiceData(
            rb.out_indices, indices, rb.pair_fwd,
            rb.pair_bwd if rb.pair_bwd is not None else torch.Tensor(),
            pair_mask_fwd_splits=[rb.mask_fwd],
            pair_mask_bwd_splits=[] if self.subm else [rb.mask_bwd],
            mask_argsort_fwd_splits=[rb.argsort_fwd] if rb.argsort_fwd is not None else [],
            mask_argsort_bwd_splits=[rb.argsort_bwd] if rb.argsort_bwd is not None else [],
            masks=masks, spatial_shape=spatial_shape, out_spatial_shape=out_spatial_shape,
            **common)

    def _conv_forward(self, training: bool, input: SparseConvTensor, weight: torch.Tensor,
                      bias: Optional[torch.Tensor], add_input: Optional[SparseConvTensor] = None,
                      channel_scale: Optional[torch.Tensor] = None,
                      output_scale: Optional[float] = None, name: Optional[str] = None,
                      sparse_unique_name: str = "", act_type=Activation.None_,
                      act_alpha: float = 0, act_beta: float = 0):
        assert isinstance(input, SparseConvTensor)
        is_int8 = input.is_quantized and weight.is_quantized
        if is_int8:
            assert output_scale is not None and channel_scale is not None, \
                "int8 must be called in static quantized module"
            assert bias is not None, "currently you must specify a bias"
            assert not training, "int8 is inference only"
            assert not self.inverse, "int8 supports regular and subm convolutions"
        assert input.features.shape[1] == self.in_channels, "channel size mismatch"
        features = input.features
        indices = input.indices
        spatial_shape = input.spatial_shape
        batch_size = input.batch_size
        # the differentiable path: training, and -- for ConvAlgo.Native layers, whose reference branch
        # always goes through the autograd functions (conv.py:327-339) -- evaluation mode with
        # gradients flowing (frozen-BN fine-tuning, saliency maps); a fused activation is then applied
        # outside.  The implicit-GEMM algos follow `training`, like the reference (conv.py:404-452).
        grad_path = training or (self.algo == ConvAlgo.Native and torch.is_grad_enabled() and not is_int8
                                 and (input.features.requires_grad or weight.requires_grad))
        bias_for_training = bias if grad_path else None
        bias_for_infer = bias if not grad_path else None
        if training:
            assert self.act_type == Activation.None_, \
                "act don't support backward, only used in inference"
        if self.subm:
            out_spatial_shape = spatial_shape
        elif self.transposed:
            out_spatial_shape = ops.get_deconv_output_size(spatial_shape, self.kernel_size,
                                                           self.stride, self.padding,
                                                           self.dilation, self.output_padding)
        else:
            out_spatial_shape = ops.get_conv_output_size(spatial_shape, self.kernel_size,
                                                         self.stride, self.padding, self.dilation)
        out_tensor = input.shadow_copy()
        if input.benchmark:
            if name is None:
                raise ValueError("you need to assign name to spmodules before benchmark "
                                 "(spconv.utils.bench.assign_name_to_spmod)")
            if name not in input.benchmark_record:
                input.benchmark_record[name] = {
                    "type": "SparseConvolution", "indice_gen_time": [], "time": [],
                    "num_points": [], "num_out_points": [],
                    "params": {"kernel_size": self.kernel_size, "stride": self.stride,
                               "padding": self.padding, "dilation": self.dilation,
                               "output_padding": self.output_padding, "subm": self.subm,
                               "transposed": self.transposed,
                               "input_channels": self.in_channels,
                               "out_channels": self.out_channels}}
        if self.conv1x1 and not is_int8:      # (int8 1x1 layers take the kernel, like the reference: conv.py:225)
            # reference quirk kept on purpose (conv.py:232-234): the [K,1..,C] weight is
            # *viewed* as [C, K] without a transpose.
            features = torch.mm(input.features, weight.view(self.in_channels, self.out_channels))
            if bias is not None:
                features += bias
            out_tensor = out_tensor.replace_feature(features)
            out_tensor.spatial_shape = out_spatial_shape
            return out_tensor
        indice_dict = input.indice_dict.copy()
        if not features.is_contiguous():
            features = features.contiguous()
        algo = self.algo
        datas = input.find_indice_pair(self.indice_key)
        if self.indice_key is not None and datas is not None:
            msg = ("due to limitation of pytorch, you must provide same algo to layers share "
                   "same indice key.")
            assert algo == datas.algo, msg
        if datas is not None:
            assert isinstance(datas, IndiceData if algo == ConvAlgo.Native
                              else ImplicitGemmIndiceData)

        if self.inverse:
            assert datas is not None and self.indice_key is not None
            assert datas.is_subm is False, \
                "inverse conv can only be used with standard conv and pool ops."
            self._check_inverse_reuse_valid(input, spatial_shape, datas)
            rb: Rulebook = datas.rulebook
            outids = datas.indices
            out_spatial_shape = datas.spatial_shape
        elif self.indice_key is not None and datas is not None:
            assert self.subm, "only support reuse subm indices"
            self._check_subm_reuse_valid(input, spatial_shape, datas)
            rb = datas.rulebook
            outids = datas.out_indices
            # The layer that BUILT this rulebook decided from its own shapes whether the Native lists (and the wgrad range
            # plan) are needed; a later layer under the same key can need them when the builder did not (other channel
            # counts).  Derived here, in the forward, not inside this layer's first backward -- where the extra launches
            # would land in a timed or captured backward pass (round-4 ADVICE).
            if (not rb.has_native and torch.is_grad_enabled() and (features.requires_grad or self.weight.requires_grad)
                    and self._needs_native_lists(features.dtype, indices, batch_size, spatial_shape)):
                rb._ensure_native()
                ops._plan_of(rb)
        else:
            if input.benchmark:
                torch.cuda.synchronize()
                t = time.time()
            try:
                with _timed(input, sparse_unique_name or name, "gen_pairs"):
                    # (a rulebook the container built ahead of this layer on its side stream, or the build itself)
                    rb = prefetch.take(self, indices, batch_size, spatial_shape)
                    if rb is None:
                        rb = self._build_rulebook(indices, batch_size, spatial_shape, features.dtype,
                                                  getattr(input, "n_live_dev", None))
                self._static_n_out_dev = rb.n_out_dev
            except Exception:
                # reference conv.py:289-297: say what was asked for and keep the inputs for a report
                print(f"[Exception|rulebook] indices={tuple(indices.shape)},bs={batch_size},ss={spatial_shape},"
                      f"algo={algo},ksize={self.kernel_size},stride={self.stride},padding={self.padding},"
                      f"dilation={self.dilation},subm={self.subm},transpose={self.transposed}", file=sys.stderr)
                save_debug_data((indices.detach().cpu().numpy(), batch_size, spatial_shape, self.kernel_size,
                                 self.stride, self.padding, self.dilation, self.subm, self.transposed))
                raise
            if input.benchmark:
                torch.cuda.synchronize()
                out_tensor.benchmark_record[name]["indice_gen_time"].append(time.time() - t)
            outids = rb.out_indices
            if self.indice_key is not None:
                msg = f"your indice key {self.indice_key} already exists in this sparse tensor."
                assert self.indice_key not in indice_dict, msg
                indice_dict[self.indice_key] = self._make_indice_data(
                    rb, indices, spatial_shape, out_spatial_shape, algo)
        if input.benchmark:
            torch.cuda.synchronize()
            t = time.time()

        num_out = outids.shape[0]
        sink = ops.current_stats_sink()
        if sink is not None:       # (BatchNorm statistics out of this layer's epilogue: the live rows of ITS output)
            sink.n_live = (rb.in_n_live_dev if self.inverse else
                           getattr(input, "n_live_dev", None) if self.subm else rb.out_n_live_dev)
        with _timed(input, sparse_unique_name or name, "forward"):
            out_features = self._run_kernels(grad_path, is_int8, input, features, weight, rb, num_out, algo,
                                             bias_for_infer, act_type, act_alpha, act_beta, output_scale,
                                             channel_scale, add_input)
        if bias_for_training is not None:
            out_features += bias_for_training
        if grad_path and not training and add_input is None:
            out_features = _apply_act(out_features, act_type, act_alpha, act_beta)
        out = self._finish(input, out_tensor, out_features, outids, indice_dict, out_spatial_shape, add_input,
                           is_int8, name, features, t if input.benchmark else None, rb)
        # live rows of a static-shape tensor: SubM keeps the input's, a strided layer has its own count, an
        # inverse layer returns to the rows its partner started from
        out.n_live_dev = (rb.in_n_live_dev if self.inverse else
                          getattr(input, "n_live_dev", None) if self.subm else rb.out_n_live_dev)
        return out

    def _build_rulebook(self, indices, batch_size, spatial_shape, feat_dtype, n_live_dev) -> Rulebook:
        """The rulebook this layer builds when no layer before it left one under its `indice_key` (also called ahead of
        the layer, on a side stream, by spconv_amd.pytorch.prefetch).  static shapes (spconv_amd.pytorch.static): a
        strided layer with a frozen output bound builds its rulebook without the device -> host read of the output
        count -- in training mode too (the Native lists come out of the same build; dead rows are in no pair)."""
        static = 0 if self.subm else int(getattr(self, "static_num_out", 0) or 0)
        rb, _ = ops.build_rulebook(indices, batch_size, spatial_shape, self.kernel_size,
                                   self.stride, self.padding, self.dilation,
                                   self.output_padding, self.subm, self.transposed,
                                   do_sort=False if static else MODULE_DO_SORT,
                                   need_native=self._needs_native_lists(feat_dtype, indices, batch_size, spatial_shape),
                                   static_num_out=static, pred_key=self,
                                   out_order=constants.CONV_OUTPUT_ORDER)
        rb.in_n_live_dev = n_live_dev
        if rb.n_out_dev is not None and getattr(rb, "out_n_live_dev", None) is None:
            rb.out_n_live_dev = rb.n_out_dev[:1].clamp(max=rb.n_out)      # live output rows: found, at most the bound
        return rb

    def _needs_native_lists(self, feat_dtype, indices, batch_size, spatial_shape) -> bool:
        """The ConvAlgo.Native lists (and the range plan built from them) feed the pair-list weight gradient; an
        inference pass does not need them, and neither does a SubM layer whose backward takes the rows walk (16 / 32
        channels on a dense level: ops.rows_backward_expected).  Left out of the build, they are derived from the
        table if something asks for them later (Rulebook.pair_native)."""
        if self.algo == ConvAlgo.Native:           # the lists ARE this algorithm's rulebook (IndiceData)
            return True
        if not torch.is_grad_enabled():
            return False
        if self.subm and not self.inverse:
            kv = int(np.prod(self.kernel_size))
            if ops.rows_backward_expected(feat_dtype, self.in_channels, self.out_channels, kv, indices.shape[0],
                                          batch_size, spatial_shape):
                return False
        return True

    def _run_kernels(self, grad_path, is_int8, input, features, weight, rb, num_out, algo, bias_for_infer,
                     act_type, act_alpha, act_beta, output_scale, channel_scale, add_input):
        if grad_path:
            # autograd path; bias is added outside the kernel like the reference
            pair_native, pair_num = (ops.attach_rulebook(rb.pair_native, rb), rb.num_per_loc) if self.inverse \
                else rb.native_handle()
            fn = (Fsp.indice_subm_conv if self.subm
                  else Fsp.indice_inverse_conv if self.inverse else Fsp.indice_conv)
            out_features = fn(features, weight, pair_native, pair_num, num_out, algo)
        elif is_int8:
            # quantised inference (conv.py:463-490): add + activation are fused in the kernel
            out_features, _, _ = ops.implicit_gemm(
                features, weight, ops.attach_rulebook(rb.pair_fwd, rb), [rb.mask_fwd],
                [rb.argsort_fwd] if rb.argsort_fwd is not None else [], num_out, [], False,
                self.subm, None, self.fp32_accum, bias_for_infer, act_alpha, act_beta, act_type,
                output_scale, channel_scale,
                output_add=add_input.features if add_input is not None else None,
                output_add_scale=add_input.q_scale() if add_input is not None else 0.0)
        else:
            w = weight if weight.dtype == features.dtype else weight.to(features.dtype)
            which = "bwd" if self.inverse else "fwd"
            table, mask, argsort, tile_order = ops.tables_of(rb, which, w.shape[0])
            ident = rb.kv // 2 if (self.subm and not self.inverse) else -1
            out_features = ops.igemm_fwd(features, w, table, mask, argsort, num_out, ident,
                                         bias_for_infer, act_type, act_alpha, tile_order=tile_order)
        return out_features

    def _finish(self, input, out_tensor, out_features, outids, indice_dict, out_spatial_shape, add_input,
                is_int8, name, features, t0, rb):
        if t0 is not None:
            torch.cuda.synchronize()
            out_tensor.benchmark_record[name]["time"].append(time.time() - t0)
            out_tensor.benchmark_record[name]["num_points"].append(features.shape[0])
            out_tensor.benchmark_record[name]["num_out_points"].append(out_features.shape[0])
        if not self.subm and not self.inverse and self.record_voxel_count:
            buf = getattr(self, _MAX_NUM_VOXELS_DURING_TRAINING, None)
            if buf is not None:
                ops.record_voxel_count_(buf, rb, outids.shape[0])
        out_tensor = out_tensor.replace_feature(out_features)
        out_tensor.indices = outids
        out_tensor.indice_dict = indice_dict
        out_tensor.spatial_shape = out_spatial_shape
        if add_input is not None and not is_int8:   # in int8 add + act happen in the kernel
            out_tensor = out_tensor.replace_feature(
                _apply_act(out_tensor.features + add_input.features, self.act_type,
                           self.act_alpha, self.act_beta))
        return out_tensor


def _timed(input: SparseConvTensor, layer: Optional[str], what: str):
    """Timer context of the tensor's CUDAKernelTimer ("<layer>.<what>"), or a no-op."""
    timer = input._timer
    if timer is None or not timer.enable:
        return contextlib.nullcontext()
    stack = contextlib.ExitStack()
    stack.enter_context(timer.namespace(layer or "conv"))
    stack.enter_context(timer.record(what))
    return stack


def _conv_cls(ndim: int, doc: str):
    class _Conv(SparseConvolution):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                     dilation=1, groups=1, bias=True, indice_key=None,
                     algo: Optional[ConvAlgo] = None, fp32_accum: Optional[bool] = None,
                     record_voxel_count: bool = False, large_kernel_fast_algo: bool = False,
                     name=None):
            super().__init__(ndim, in_channels, out_channels, kernel_size, stride, padding,
                             dilation, groups, bias, indice_key=indice_key, algo=algo,
                             fp32_accum=fp32_accum, record_voxel_count=record_voxel_count,
                             large_kernel_fast_algo=large_kernel_fast_algo, name=name)
    _Conv.__doc__ = doc
    return _Conv


def _deconv_cls(ndim: int, doc: str):
    class _ConvT(SparseConvolution):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                     dilation=1, groups=1, bias=True, indice_key=None,
                     algo: Optional[ConvAlgo] = None, fp32_accum: Optional[bool] = None,
                     record_voxel_count: bool = False, large_kernel_fast_algo: bool = False,
                     name=None):
            super().__init__(ndim, in_channels, out_channels, kernel_size, stride, padding,
                             dilation, groups, bias, transposed=True, indice_key=indice_key,
                             algo=algo, fp32_accum=fp32_accum,
                             record_voxel_count=record_voxel_count,
                             large_kernel_fast_algo=large_kernel_fast_algo, name=name)
    _ConvT.__doc__ = doc
    return _ConvT


def _inverse_cls(ndim: int, doc: str):
    class _Inv(SparseConvolution):
        def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True,
                     algo: Optional[ConvAlgo] = None, fp32_accum: Optional[bool] = None,
                     large_kernel_fast_algo: bool = False, name=None):
            super().__init__(ndim, in_channels, out_channels, kernel_size, bias=bias,
                             inverse=True, indice_key=indice_key, algo=algo,
                             fp32_accum=fp32_accum,
                             large_kernel_fast_algo=large_kernel_fast_algo, name=name)
    _Inv.__doc__ = doc
    return _Inv


def _subm_cls(ndim: int, doc: str):
    class _SubM(SparseConvolution):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                     dilation=1, groups=1, bias=True, indice_key=None,
                     algo: Optional[ConvAlgo] = None, fp32_accum: Optional[bool] = None,
                     large_kernel_fast_algo: bool = False, name=None):
            super().__init__(ndim, in_channels, out_channels, kernel_size, stride, padding,
                             dilation, groups, bias, True, indice_key=indice_key, algo=algo,
                             fp32_accum=fp32_accum,
                             large_kernel_fast_algo=large_kernel_fast_algo, name=name)
    _SubM.__doc__ = doc
    return _SubM


def _named(cls, name):
    cls.__name__ = cls.__qualname__ = name
    return cls


for _n in (1, 2, 3, 4):
    globals()[f"SparseConv{_n}d"] = _named(
        _conv_cls(_n, f"{_n}-d sparse convolution (reference conv.py:767-905)."), f"SparseConv{_n}d")
    globals()[f"SparseConvTranspose{_n}d"] = _named(
        _deconv_cls(_n, f"{_n}-d sparse transposed convolution (reference conv.py:907-1049)."),
        f"SparseConvTranspose{_n}d")
    globals()[f"SparseInverseConv{_n}d"] = _named(
        _inverse_cls(_n, f"{_n}-d inverse of a SparseConv sharing indice_key "
                         f"(reference conv.py:1051-1153)."), f"SparseInverseConv{_n}d")
    globals()[f"SubMConv{_n}d"] = _named(
        _subm_cls(_n, f"{_n}-d submanifold convolution (reference conv.py:1155-1308)."),
        f"SubMConv{_n}d")


# reference conv.py:1291-1308: the concrete layer classes quantization patterns are keyed on
DEFAULT_SPARSE_CONV_TYPES = {
    globals()[f"{_kind}{_n}d"]
    for _kind in ("SubMConv", "SparseConv", "SparseInverseConv", "SparseConvTranspose")
    for _n in (1, 2, 3, 4)
}


def conv_ctor_kwargs(conv: "SparseConvolution") -> dict:
    """Constructor arguments that rebuild ``conv`` as a plain SparseConvolution (or a subclass
    taking the same arguments): used by the quantization modules' from_float / to_float."""
    return dict(ndim=conv.ndim, in_channels=conv.in_channels, out_channels=conv.out_channels,
                kernel_size=conv.kernel_size, stride=conv.stride, padding=conv.padding,
                dilation=conv.dilation, groups=conv.groups, bias=conv.bias is not None
                if not callable(conv.bias) else True,
                subm=conv.subm, output_padding=conv.output_padding, transposed=conv.transposed,
                inverse=conv.inverse, indice_key=conv.indice_key, algo=conv.algo,
                fp32_accum=conv.fp32_accum, record_voxel_count=conv.record_voxel_count,
                act_type=conv.act_type, act_alpha=conv.act_alpha, act_beta=conv.act_beta,
                name=conv.name)
