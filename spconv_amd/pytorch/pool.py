"""Sparse pooling modules (reference: ``spconv/pytorch/pool.py:36-505``): ``SparseMaxPool{1..4}d``,
``SparseAvgPool{1..3}d``, ``SparseGlobalMaxPool`` / ``SparseGlobalAvgPool``.  Same constructor
signatures; the rulebook is the regular-conv rulebook of the pooling window (or the SubM one with
``subm=True``), built by the same HIP builder as for convolutions and registered under
``indice_key`` so that a ``SparseInverseConv`` can undo the pooling."""
from __future__ import annotations

import time
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from spconv_amd.pytorch import functional as Fsp
from spconv_amd import constants
from spconv_amd.pytorch import ops
from spconv_amd.pytorch.core import (ConvAlgo, ImplicitGemmIndiceData, IndiceData, SparseConvTensor,
                                     expand_nd)
from spconv_amd.pytorch.modules import SparseModule

_MAX_NUM_VOXELS_DURING_TRAINING = "max_num_voxels_during_training"


class _SparsePoolBase(SparseModule):
    _is_avg = False

    def __init__(self, ndim, kernel_size: Union[int, List[int], Tuple[int, ...]] = 3,
                 stride: Optional[Union[int, List[int], Tuple[int, ...]]] = 1,
                 padding: Union[int, List[int], Tuple[int, ...]] = 0,
                 dilation: Union[int, List[int], Tuple[int, ...]] = 1,
                 indice_key: Optional[str] = None, subm: bool = False,
                 algo: Optional[ConvAlgo] = None, record_voxel_count: bool = False, name=None):
        super().__init__(name=name)
        self.ndim = ndim
        self.kernel_size = expand_nd(ndim, kernel_size)
        self.stride = self.kernel_size.copy() if stride is None else expand_nd(ndim, stride)
        self.padding = expand_nd(ndim, padding)
        self.dilation = expand_nd(ndim, dilation)
        self.subm = subm
        self.indice_key = indice_key
        self.record_voxel_count = record_voxel_count
        if record_voxel_count and not self.subm:
            self.register_buffer(_MAX_NUM_VOXELS_DURING_TRAINING, torch.zeros(1, dtype=torch.int32))
        kv = int(np.prod(self.kernel_size))
        if self._is_avg:
            assert kv <= 32, "avg pool only support implicit-gemm style indice gen with kv <= 32 limit"
            algo = ConvAlgo.MaskImplicitGemm
        elif algo is None:
            algo = ConvAlgo.MaskImplicitGemm if kv <= 128 else ConvAlgo.Native
        self.algo = algo

    def extra_repr(self):
        s = "kernel_size={kernel_size}, stride={stride}"
        if self.padding != [0] * len(self.padding):
            s += ", padding={padding}"
        if self.dilation != [1] * len(self.dilation):
            s += ", dilation={dilation}"
        if self.algo is not None:
            s += f", algo={self.algo}"
        return s.format(**self.__dict__)

    def get_max_num_voxels(self) -> Optional[torch.Tensor]:
        return getattr(self, _MAX_NUM_VOXELS_DURING_TRAINING, None)

    def forward(self, input: SparseConvTensor):
        assert isinstance(input, SparseConvTensor)
        if input.is_quantized:
            assert not self._is_avg and self.algo == ConvAlgo.MaskImplicitGemm, \
                "only max pooling with ConvAlgo.MaskImplicitGemm supports int8"
        features, indices = input.features, input.indices
        spatial_shape, batch_size = input.spatial_shape, input.batch_size
        out_spatial_shape = spatial_shape if self.subm else ops.get_conv_output_size(
            spatial_shape, self.kernel_size, self.stride, self.padding, self.dilation)
        out_tensor = input.shadow_copy()
        indice_dict = input.indice_dict.copy()
        if input.benchmark:
            if self.name is None:
                raise ValueError("you need to assign name to spmodules before benchmark "
                                 "(spconv.utils.bench.assign_name_to_spmod)")
            input.benchmark_record.setdefault(self.name, {
                "type": type(self).__name__, "indice_gen_time": [], "time": [], "num_points": [],
                "num_out_points": [],
                "params": {"kernel_size": self.kernel_size, "stride": self.stride,
                           "padding": self.padding, "dilation": self.dilation,
                           "channels": features.shape[1]}})
            torch.cuda.synchronize()
            t = time.time()
        if self.indice_key is not None and input.find_indice_pair(self.indice_key) is not None:
            raise ValueError(f"indice key {self.indice_key} exists")
        # static shapes (spconv_amd.pytorch.static): frozen output bound, no read-back -- inference and training
        # (the reference's bounded mode covers pools too: pool.py:99-248 via ops.py:263-266).  The table-driven
        # backward kernels skip dead rows by construction (no pair); ConvAlgo.Native takes its lists from the same
        # sync-free build.
        static = 0 if self.subm else int(getattr(self, "static_num_out", 0) or 0)
        rb, _ = ops.build_rulebook(indices, batch_size, spatial_shape, self.kernel_size, self.stride,
                                   self.padding, self.dilation, [0] * self.ndim, self.subm, False,
                                   need_bwd_table=True,
                                   need_native=(not static) or self.algo == ConvAlgo.Native,
                                   static_num_out=static, out_order=constants.CONV_OUTPUT_ORDER)
        self._static_n_out_dev = rb.n_out_dev
        rb.in_n_live_dev = getattr(input, "n_live_dev", None)
        if rb.n_out_dev is not None and getattr(rb, "out_n_live_dev", None) is None:
            rb.out_n_live_dev = rb.n_out_dev[:1].clamp(max=rb.n_out)
        outids = rb.out_indices
        if input.benchmark:
            torch.cuda.synchronize()
            out_tensor.benchmark_record[self.name]["indice_gen_time"].append(time.time() - t)
            t = time.time()
        common = dict(is_subm=self.subm, algo=self.algo, ksize=self.kernel_size, stride=self.stride,
                      dilation=self.dilation, padding=self.padding, rulebook=rb)
        if self.algo == ConvAlgo.Native:
            if self.indice_key is not None:
                indice_dict[self.indice_key] = IndiceData(outids, indices, rb.pair_native, rb.num_per_loc,
                                                          spatial_shape, out_spatial_shape, **common)
            out_features = Fsp.indice_maxpool(features, ops.attach_rulebook(rb.pair_native, rb),
                                              rb.num_per_loc, outids.shape[0])
        else:
            if self.indice_key is not None:
                indice_dict[self.indice_key] = ImplicitGemmIndiceData(
                    outids, indices, rb.pair_fwd, rb.pair_bwd, pair_mask_fwd_splits=[rb.mask_fwd],
                    pair_mask_bwd_splits=[rb.mask_bwd], mask_argsort_fwd_splits=[],
                    mask_argsort_bwd_splits=[], masks=[np.array([0xffffffff], dtype=np.uint32)],
                    spatial_shape=spatial_shape, out_spatial_shape=out_spatial_shape, **common)
            pair_fwd = ops.attach_rulebook(rb.pair_fwd, rb)
            pair_bwd = ops.attach_rulebook(rb.pair_bwd, rb)
            if self._is_avg:
                out_features = Fsp.indice_avgpool_implicit_gemm(features, pair_fwd, pair_bwd,
                                                                outids.shape[0], self.training)
            else:
                out_features = Fsp.indice_maxpool_implicit_gemm(features, pair_fwd, pair_bwd,
                                                                outids.shape[0])
        if input.benchmark:
            torch.cuda.synchronize()
            rec = out_tensor.benchmark_record[self.name]
            rec["time"].append(time.time() - t)
            rec["num_points"].append(features.shape[0])
            rec["num_out_points"].append(out_features.shape[0])
        if not self.subm and self.record_voxel_count:
            buf = getattr(self, _MAX_NUM_VOXELS_DURING_TRAINING, None)
            if buf is not None:
                ops.record_voxel_count_(buf, rb, outids.shape[0])
        out_tensor = out_tensor.replace_feature(out_features)
        out_tensor.indices = outids
        out_tensor.indice_dict = indice_dict
        out_tensor.spatial_shape = list(out_spatial_shape)
        out_tensor.n_live_dev = getattr(input, "n_live_dev", None) if self.subm else rb.out_n_live_dev
        return out_tensor


class SparseMaxPool(_SparsePoolBase):
    """reference pool.py:36-253"""


class SparseAvgPool(_SparsePoolBase):
    """reference pool.py:294-432"""
    _is_avg = True


class SparseGlobalMaxOrAvgPool(SparseModule):
    """Per-scene reduction over all voxels (reference pool.py:255-292): returns a dense
    ``[batch_size, C]`` tensor; autograd comes from the torch reductions."""

    def __init__(self, is_mean: bool, name=None):
        super().__init__(name=name)
        self.is_mean = is_mean

    def forward(self, input: SparseConvTensor):
        assert isinstance(input, SparseConvTensor) and not input.is_quantized, "not implemented"
        # one segmented reduction over the batch column (no per-scene loop, no read-back); rows whose
        # batch index lies outside [0, batch_size) belong to no scene, as in global_pool_rearrange
        bs, feats = input.batch_size, input.features
        b = input.indices[:, 0].long()
        keep = (b >= 0) & (b < bs)
        b = torch.where(keep, b, torch.zeros_like(b))
        idx = b.unsqueeze(1).expand(-1, feats.shape[1])
        if self.is_mean:
            # sums and row counts in fp32 whatever the feature dtype (torch.mean of the reference loop
            # accumulates in fp32 too): an fp16 count saturates at 2048 rows, a bf16 one at 256, and a scene
            # has tens of thousands.  A scene without rows gives NaN, like torch.mean over no rows.
            acc_t = torch.float64 if feats.dtype == torch.float64 else torch.float32
            w = keep.to(acc_t).unsqueeze(1)
            total = torch.zeros((bs, feats.shape[1]), dtype=acc_t, device=feats.device).scatter_add(
                0, idx, feats.to(acc_t) * w)
            count = torch.zeros((bs, 1), dtype=acc_t, device=feats.device).scatter_add(0, b.unsqueeze(1), w)
            return (total / count).to(feats.dtype)
        lowest = torch.finfo(feats.dtype).min if feats.dtype.is_floating_point else torch.iinfo(feats.dtype).min
        src = torch.where(keep.unsqueeze(1), feats, torch.full_like(feats, lowest))
        return torch.full((bs, feats.shape[1]), lowest, dtype=feats.dtype, device=feats.device).scatter_reduce(
            0, idx, src, "amax", include_self=True)


class SparseGlobalAvgPool(SparseGlobalMaxOrAvgPool):
    def __init__(self, name=None):
        super().__init__(is_mean=True, name=name)


class SparseGlobalMaxPool(SparseGlobalMaxOrAvgPool):
    def __init__(self, name=None):
        super().__init__(is_mean=False, name=name)


def _pool_cls(base, ndim: int, name: str):
    def __init__(self, kernel_size, stride=None, padding=0, dilation=1, indice_key=None,
                 algo: Optional[ConvAlgo] = None, record_voxel_count: bool = False, name=None):
        base.__init__(self, ndim, kernel_size, stride, padding, dilation, indice_key=indice_key,
                      algo=algo, record_voxel_count=record_voxel_count, name=name)
    return type(name, (base,), {"__init__": __init__,
                                "__doc__": f"{ndim}-d {base.__name__} (reference pool.py:434-505)."})


SparseMaxPool1d = _pool_cls(SparseMaxPool, 1, "SparseMaxPool1d")
SparseMaxPool2d = _pool_cls(SparseMaxPool, 2, "SparseMaxPool2d")
SparseMaxPool3d = _pool_cls(SparseMaxPool, 3, "SparseMaxPool3d")
SparseMaxPool4d = _pool_cls(SparseMaxPool, 4, "SparseMaxPool4d")
SparseAvgPool1d = _pool_cls(SparseAvgPool, 1, "SparseAvgPool1d")
SparseAvgPool2d = _pool_cls(SparseAvgPool, 2, "SparseAvgPool2d")
SparseAvgPool3d = _pool_cls(SparseAvgPool, 3, "SparseAvgPool3d")
