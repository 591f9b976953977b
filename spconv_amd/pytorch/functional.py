"""autograd glue (reference: spconv/pytorch/functional.py:59-429).

``SparseConvFunction`` / ``SparseInverseConvFunction`` / ``SubMConvFunction`` /
``SparseImplicitGemmFunction`` keep the reference's argument lists so user code
calling ``Fsp.indice_subm_conv(...)`` etc. keeps working; all of them save
(features, filters, rulebook tensors) and return ``(din, dW, None...)``.
AMP: inputs are cast to fp16 under autocast like the reference
(functional.py:44-56).
"""
from __future__ import annotations

import sys
from typing import List, Optional

import numpy as np
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from spconv_amd.pytorch import ops

import torch.amp as _amp

_FWD = _amp.custom_fwd(device_type="cuda", cast_inputs=torch.float16)
_BWD = _amp.custom_bwd(device_type="cuda")


def _report(tag: str, **shapes) -> None:
    msg = f"[Exception|{tag}]" + ",".join(f"{k}={v}" for k, v in shapes.items())
    print(msg, file=sys.stderr)


class _NativeConvBase(Function):
    """Shared body of the three ConvAlgo.Native functions."""
    _inverse = False
    _subm = False

    @classmethod
    def _forward(cls, ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out,
                 algo, timer, bias, act_alpha, act_beta, act_type):
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, filters)
        ctx.algo = algo
        # tensors lose python attributes through save_for_backward; keep the rulebook here
        ctx.rulebook = ops.rulebook_of(indice_pairs)
        try:
            return ops.indice_conv(features, filters, indice_pairs, indice_pair_num,
                                   num_activate_out, cls._inverse, cls._subm, algo=algo,
                                   timer=timer, bias=bias, act_alpha=act_alpha,
                                   act_beta=act_beta, act_type=act_type)
        except Exception:
            _report("indice_conv", feat=features.shape, w=filters.shape, pair=indice_pairs.shape,
                    act=num_activate_out, algo=algo)
            raise

    @classmethod
    def _backward(cls, ctx, grad_output):
        indice_pairs, indice_pair_num, features, filters = ctx.saved_tensors
        if ctx.rulebook is not None:
            ops.attach_rulebook(indice_pairs, ctx.rulebook)
        try:
            din, dw = ops.indice_conv_backward(features, filters, grad_output, indice_pairs,
                                               indice_pair_num, cls._inverse, cls._subm,
                                               algo=ctx.algo, need_din=ctx.needs_input_grad[0])
        except Exception:
            _report("indice_conv_backward", feat=features.shape, w=filters.shape,
                    pair=indice_pairs.shape, do=grad_output.shape)
            raise
        return (din, dw) + (None,) * 9


class SparseConvFunction(_NativeConvBase):
    @staticmethod
    @_FWD
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out, algo,
                timer=None, bias: Optional[torch.Tensor] = None, act_alpha: float = 0.0,
                act_beta: float = 0.0, act_type=ops.Activation.None_):
        return SparseConvFunction._forward(ctx, features, filters, indice_pairs, indice_pair_num,
                                           num_activate_out, algo, timer, bias, act_alpha,
                                           act_beta, act_type)

    @staticmethod
    @once_differentiable
    @_BWD
    def backward(ctx, grad_output):
        return SparseConvFunction._backward(ctx, grad_output)


class SparseInverseConvFunction(_NativeConvBase):
    _inverse = True

    @staticmethod
    @_FWD
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out, algo,
                timer=None, bias: Optional[torch.Tensor] = None, act_alpha: float = 0.0,
                act_beta: float = 0.0, act_type=ops.Activation.None_):
        return SparseInverseConvFunction._forward(ctx, features, filters, indice_pairs,
                                                  indice_pair_num, num_activate_out, algo, timer,
                                                  bias, act_alpha, act_beta, act_type)

    @staticmethod
    @once_differentiable
    @_BWD
    def backward(ctx, grad_output):
        return SparseInverseConvFunction._backward(ctx, grad_output)


class SubMConvFunction(_NativeConvBase):
    _subm = True

    @staticmethod
    @_FWD
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out, algo,
                timer=None, bias: Optional[torch.Tensor] = None, act_alpha: float = 0.0,
                act_beta: float = 0.0, act_type=ops.Activation.None_):
        return SubMConvFunction._forward(ctx, features, filters, indice_pairs, indice_pair_num,
                                         num_activate_out, algo, timer, bias, act_alpha, act_beta,
                                         act_type)

    @staticmethod
    @once_differentiable
    @_BWD
    def backward(ctx, grad_output):
        return SubMConvFunction._backward(ctx, grad_output)


class SparseImplicitGemmFunction(Function):
    @staticmethod
    @_FWD
    def forward(ctx, features: torch.Tensor, filters: torch.Tensor, pair_fwd: torch.Tensor,
                pair_bwd: torch.Tensor, pair_mask_fwd_splits: List[torch.Tensor],
                pair_mask_bwd_splits: List[torch.Tensor],
                mask_argsort_fwd_splits: List[torch.Tensor],
                mask_argsort_bwd_splits: List[torch.Tensor], num_activate_out: int,
                masks: List[np.ndarray], is_train: bool, is_subm: bool, timer=None,
                fp32_accum: Optional[bool] = None, bias: Optional[torch.Tensor] = None,
                act_alpha: float = 0.0, act_beta: float = 0.0,
                act_type=ops.Activation.None_):
        try:
            out, mask_out, mask_width = ops.implicit_gemm(
                features, filters, pair_fwd, pair_mask_fwd_splits, mask_argsort_fwd_splits,
                num_activate_out, masks, is_train, is_subm, timer, fp32_accum, bias, act_alpha,
                act_beta, act_type)
        except Exception:
            _report("implicit_gemm", feat=features.shape, w=filters.shape, pair=pair_fwd.shape,
                    act=num_activate_out, issubm=is_subm, istrain=is_train)
            raise
        ctx.save_for_backward(features, filters, pair_fwd, pair_bwd)
        ctx.rulebook = ops.rulebook_of(pair_fwd)
        ctx.mask_width = mask_width
        ctx.mask_out = mask_out
        ctx.pair_mask_fwd_splits = pair_mask_fwd_splits
        ctx.mask_argsort_fwd_splits = mask_argsort_fwd_splits
        ctx.pair_mask_bwd_splits = pair_mask_bwd_splits
        ctx.mask_argsort_bwd_splits = mask_argsort_bwd_splits
        ctx.masks = masks
        ctx.is_subm = is_subm
        ctx.fp32_accum = fp32_accum
        return out

    @staticmethod
    @once_differentiable
    @_BWD
    def backward(ctx, grad_output):
        features, filters, pair_fwd, pair_bwd = ctx.saved_tensors
        if ctx.rulebook is not None:
            ops.attach_rulebook(pair_fwd, ctx.rulebook)
        try:
            din, dw = ops.implicit_gemm_backward(
                features, filters, grad_output, pair_fwd, pair_bwd, ctx.pair_mask_fwd_splits,
                ctx.pair_mask_bwd_splits, ctx.mask_argsort_fwd_splits,
                ctx.mask_argsort_bwd_splits, mask_output_fwd=ctx.mask_out, masks=ctx.masks,
                mask_width=ctx.mask_width, is_subm=ctx.is_subm, fp32_accum=ctx.fp32_accum,
                need_din=ctx.needs_input_grad[0])
        except Exception:
            _report("implicit_gemm_backward", feat=features.shape, w=filters.shape,
                    pair=pair_fwd.shape, issubm=ctx.is_subm, do=grad_output.shape)
            raise
        return (din, dw) + (None,) * 16


class SparseMaxPoolFunction(Function):
    """ConvAlgo.Native max pool (reference functional.py:360-377)."""
    @staticmethod
    @_FWD
    def forward(ctx, features, indice_pairs, indice_pair_num, num_activate_out):
        out = ops.indice_maxpool(features, indice_pairs, indice_pair_num, num_activate_out)
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, out)
        ctx.rulebook = ops.rulebook_of(indice_pairs)
        return out

    @staticmethod
    @once_differentiable
    @_BWD
    def backward(ctx, grad_output):
        indice_pairs, indice_pair_num, features, out = ctx.saved_tensors
        if ctx.rulebook is not None:
            ops.attach_rulebook(indice_pairs, ctx.rulebook)
        input_bp = ops.indice_maxpool_backward(features, out, grad_output, indice_pairs, indice_pair_num)
        return input_bp, None, None, None


class SparseMaxPoolImplicitGemmFunction(Function):
    """reference functional.py:380-397"""
    @staticmethod
    @_FWD
    def forward(ctx, features: torch.Tensor, indice_pairs_fwd: torch.Tensor,
                indice_pairs_bwd: torch.Tensor, num_activate_out: int):
        out = ops.indice_maxpool_implicit_gemm(features, indice_pairs_fwd, num_activate_out)
        ctx.save_for_backward(indice_pairs_bwd, features, out)
        ctx.rulebook = ops.rulebook_of(indice_pairs_fwd)
        return out

    @staticmethod
    @once_differentiable
    @_BWD
    def backward(ctx, grad_output):
        indice_pairs_bwd, features, out = ctx.saved_tensors
        if ctx.rulebook is not None:
            ops.attach_rulebook(indice_pairs_bwd, ctx.rulebook)
        input_bp = ops.indice_maxpool_implicit_gemm_backward(features, out, grad_output, indice_pairs_bwd)
        return input_bp, None, None, None


class SparseAvgPoolImplicitGemmFunction(Function):
    """reference functional.py:400-420"""
    @staticmethod
    @_FWD
    def forward(ctx, features: torch.Tensor, indice_pairs_fwd: torch.Tensor,
                indice_pairs_bwd: torch.Tensor, num_activate_out: int, calc_count):
        out, count = ops.indice_avgpool_implicit_gemm(features, indice_pairs_fwd, num_activate_out,
                                                      calc_count)
        ctx.save_for_backward(indice_pairs_bwd, features, out, count)
        ctx.rulebook = ops.rulebook_of(indice_pairs_fwd)
        return out

    @staticmethod
    @once_differentiable
    @_BWD
    def backward(ctx, grad_output):
        indice_pairs_bwd, features, out, count = ctx.saved_tensors
        if ctx.rulebook is not None:
            ops.attach_rulebook(indice_pairs_bwd, ctx.rulebook)
        input_bp = ops.indice_avgpool_implicit_gemm_backward(grad_output, indice_pairs_bwd, count)
        return input_bp, None, None, None, None


indice_conv = SparseConvFunction.apply
implicit_gemm = SparseImplicitGemmFunction.apply
indice_inverse_conv = SparseInverseConvFunction.apply
indice_subm_conv = SubMConvFunction.apply
indice_maxpool = SparseMaxPoolFunction.apply
indice_maxpool_implicit_gemm = SparseMaxPoolImplicitGemmFunction.apply
indice_avgpool_implicit_gemm = SparseAvgPoolImplicitGemmFunction.apply


# ------------------------------------------------------------------ sparse add
_MAX_INT32 = 2147483647


def _indice_to_scalar(indices: torch.Tensor, shape: List[int]) -> torch.Tensor:
    """Row-major linear index of (batch, *coords) rows (reference functional.py:430-436)."""
    assert indices.shape[1] == len(shape)
    out = torch.zeros_like(indices[:, 0])
    for d, extent in enumerate(shape):
        out = out * extent + indices[:, d]
    return out.contiguous()


def _like_first(first, features, indices, indice_dict=None):
    from spconv_amd.pytorch.core import SparseConvTensor
    res = SparseConvTensor(features, indices, first.spatial_shape, first.batch_size,
                           benchmark=first.benchmark)
    if indice_dict is not None:
        res.indice_dict = indice_dict
    res.benchmark_record = first.benchmark_record
    res._timer = first._timer
    res.thrust_allocator = first.thrust_allocator
    return res


def sparse_add_hash_based(*tens):
    """Sum of sparse tensors with different coordinate sets (reference functional.py:439-499):
    the union of the coordinates is numbered through a hash table.  indice_dict is dropped unless
    one operand already holds every output coordinate."""
    from spconv_amd.pytorch.hash import HashTable
    first = tens[0]
    for ten in tens:
        assert ten.spatial_shape == first.spatial_shape
        assert ten.batch_size == first.batch_size
        assert ten.features.shape[1] == first.features.shape[1]
    sizes = [t.features.shape[0] for t in tens]
    biggest = max(range(len(tens)), key=lambda i: sizes[i])
    shape = [first.batch_size, *first.spatial_shape]
    big = int(np.prod(shape)) >= _MAX_INT32
    k_type = torch.int64 if big else torch.int32
    table = HashTable(first.features.device, k_type, torch.int32, max(2 * sum(sizes), 2))
    scalars = []
    for ten in tens:
        scalar = _indice_to_scalar(ten.indices.long() if big else ten.indices, shape)
        scalars.append(scalar)
        table.insert(scalar)
    count_val = int(table.assign_arange_().item())
    feat = first.features
    out_features = torch.zeros([count_val, feat.shape[1]], dtype=feat.dtype, device=feat.device)
    out_indices = torch.zeros([count_val, first.indices.shape[1]], dtype=first.indices.dtype,
                              device=first.indices.device)
    for ten, scalar in zip(tens, scalars):
        rows = table.query(scalar)[0].long()
        out_features.index_add_(0, rows, ten.features)
        out_indices[rows] = ten.indices
    keep = tens[biggest].indice_dict if count_val == sizes[biggest] else None
    return _like_first(first, out_features, out_indices, keep)


def sparse_add(*tens):
    """Same sum through sort + unique (reference functional.py:502-545 goes through torch.sparse):
    output rows are ordered by coordinate."""
    first = tens[0]
    for ten in tens:
        assert ten.spatial_shape == first.spatial_shape
        assert ten.batch_size == first.batch_size
        assert ten.features.shape[1] == first.features.shape[1]
    sizes = [t.features.shape[0] for t in tens]
    biggest = max(range(len(tens)), key=lambda i: sizes[i])
    shape = [first.batch_size, *first.spatial_shape]
    scalars = torch.cat([_indice_to_scalar(t.indices.long(), shape) for t in tens])
    uniq, inverse = torch.unique(scalars, sorted=True, return_inverse=True)
    feats = torch.cat([t.features for t in tens])
    out_features = torch.zeros([uniq.shape[0], feats.shape[1]], dtype=feats.dtype, device=feats.device)
    out_features.index_add_(0, inverse, feats)
    cols = []
    rest = uniq
    for extent in reversed(shape):
        cols.append(rest % extent)
        rest = rest // extent
    out_indices = torch.stack(cols[::-1], dim=1).to(first.indices.dtype).contiguous()
    keep = tens[biggest].indice_dict if uniq.shape[0] == sizes[biggest] else None
    return _like_first(first, out_features, out_indices, keep)

