"""BatchNorm1d (+ ReLU) on the feature matrix of a SparseConvTensor through the HIP kernels of
csrc/norm.hip.

The reference leaves normalisation to torch: ``SparseSequential`` hands ``.features`` to whatever dense
module comes next (spconv/pytorch/modules.py:131-145) and ``SparseBatchNorm`` is ``nn.BatchNorm1d`` on
features (:147-160).  Same here -- any module still works -- but a plain ``nn.BatchNorm1d`` (and a
``nn.ReLU`` right behind it) on CUDA features of a supported shape takes this path instead of torch's
channels-last kernels, which are 4-5x slower on MI355X at backbone shapes.
Semantics follow ``torch.nn.BatchNorm1d.forward``: batch statistics when training (or when the module
keeps no running estimates), running estimates otherwise, ``momentum=None`` = cumulative average,
``num_batches_tracked`` counted.  ``SPCONV_AMD_FUSED_BN=0`` switches the path off.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch import nn

from spconv_amd import _lib

ENABLED = os.environ.get("SPCONV_AMD_FUSED_BN", "1") != "0"
_DT = {torch.float16: _lib.DTYPE_F16, torch.bfloat16: _lib.DTYPE_BF16, torch.float32: _lib.DTYPE_F32}


def supported(features: torch.Tensor, bn: nn.Module) -> bool:
    """Plain BatchNorm1d (not a subclass with its own forward, e.g. SyncBatchNorm) on a contiguous
    CUDA [N, C] matrix whose rows split into 16-byte pieces."""
    if not ENABLED or type(bn) is not nn.BatchNorm1d or not features.is_cuda or features.dim() != 2:
        return False
    if features.dtype not in _DT or bn._forward_hooks or bn._forward_pre_hooks or bn._backward_hooks:
        return False            # (user hooks fire on the module call: keep torch's path for them)
    if features.shape[0] <= 1 and (bn.training or bn.running_mean is None):
        return False            # torch raises "Expected more than 1 value per channel": keep its error
    C = features.shape[1]
    if _param_dtype(bn.weight, bn.bias, bn.running_mean, bn.running_var) not in _DT:
        return False            # parameters and buffers of mixed dtypes: torch's path
    return C == bn.num_features and C % (4 if features.dtype == torch.float32 else 8) == 0 and C <= 256


def _param_dtype(*tensors) -> Optional[torch.dtype]:
    """Common dtype of the module's parameter / buffer vectors (None when they disagree)."""
    dts = {t.dtype for t in tensors if t is not None}
    if not dts:
        return torch.float32
    return dts.pop() if len(dts) == 1 else None


class _BatchNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, relu, nbt=None,
                n_live=None, differentiable=True, records=None, n_records=0):
        # differentiable: the caller's grad mode AND an input that requires grad (decided outside: inside forward the
        # grad mode is always off, and ctx.needs_input_grad reports requires_grad whatever the mode -- an inference
        # pass under no_grad over a model whose parameters still require grad must not pay for a snapshot)
        L = _lib.load()
        x = x.contiguous()
        n, C = x.shape
        dev = x.device
        y = torch.empty_like(x)
        pdt = _param_dtype(weight, bias, running_mean, running_var)
        stats = torch.empty((2, C), dtype=torch.float32, device=dev)
        p = lambda t: None if t is None else t.data_ptr()
        if records is not None and training:
            # the statistics pass has happened already: in the epilogue of the convolution that produced x
            # (spx_igemm_fwd_stats left one {rows, mean, M2} record per workgroup) -- merge + apply, two launches
            with torch.cuda.device(dev):
                _lib.check(L.spx_batchnorm_fwd_stats(x.data_ptr(), y.data_ptr(), n, C, _DT[x.dtype], p(weight), p(bias),
                                                     p(running_mean), p(running_var), p(nbt), _DT[pdt], float(momentum),
                                                     float(eps), int(relu), stats[0].data_ptr(), stats[1].data_ptr(),
                                                     records.data_ptr(), int(n_records), p(n_live),
                                                     torch._C._cuda_getCurrentRawStream(dev.index)))
            ctx.snap = False
            ctx.save_for_backward(x, weight, bias, stats[0], stats[1])
            ctx.training, ctx.relu, ctx.pdt, ctx.n_live, ctx.eps = True, bool(relu), pdt, n_live, float(eps)
            return y
        ws = torch.empty((max(L.spx_batchnorm_ws_bytes(n, C), 16),), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.spx_batchnorm_fwd(x.data_ptr(), y.data_ptr(), n, C, _DT[x.dtype], p(weight), p(bias),
                                           p(running_mean), p(running_var), p(nbt), _DT[pdt], int(training), float(momentum),
                                           float(eps), int(relu), stats[0].data_ptr(), stats[1].data_ptr(),
                                           ws.data_ptr(), ws.numel(), p(n_live),
                                           torch._C._cuda_getCurrentRawStream(dev.index)))
        # evaluation mode: an inference pass must not pay four small launches per layer for fp32 mean / 1/std; a
        # pass that will be differentiated (frozen-BN fine-tuning) snapshots them NOW -- the kernels update the
        # running buffers through raw pointers (no autograd version bump), so a training-mode pass of the same
        # module between this forward and its backward must not change what the backward normalises with
        ctx.snap = False
        if training:
            ctx.save_for_backward(x, weight, bias, stats[0], stats[1])
        elif differentiable:
            ctx.snap = True
            ctx.save_for_backward(x, weight, bias, running_mean.float().clone(),
                                  torch.rsqrt(running_var.float() + float(eps)))
        else:
            ctx.save_for_backward(x, weight, bias, running_mean, running_var)
        ctx.training, ctx.relu, ctx.pdt, ctx.n_live, ctx.eps = bool(training), bool(relu), pdt, n_live, float(eps)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable      # no double backward: an explicit error, not silent zeros
    def backward(ctx, dy):
        L = _lib.load()
        x, weight, bias, mean, invstd = ctx.saved_tensors
        if not ctx.training and not ctx.snap:           # saved: the running estimates themselves
            mean, invstd = mean.float(), torch.rsqrt(invstd.float() + ctx.eps)
        n, C = x.shape
        dev = x.device
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dw = None if weight is None else torch.empty_like(weight)
        db = None if bias is None else torch.empty_like(bias)
        ws = torch.empty((max(L.spx_batchnorm_ws_bytes(n, C), 16),), dtype=torch.uint8, device=dev)
        p = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(L.spx_batchnorm_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), n, C, _DT[x.dtype], p(weight),
                                           p(bias), _DT[ctx.pdt], mean.data_ptr(), invstd.data_ptr(),
                                           int(ctx.training), int(ctx.relu), p(dw), p(db), ws.data_ptr(), ws.numel(),
                                           p(ctx.n_live), torch._C._cuda_getCurrentRawStream(dev.index)))
        return dx, dw, db, None, None, None, None, None, None, None, None, None, None, None


def batch_norm(features: torch.Tensor, bn: nn.BatchNorm1d, relu: bool = False,
               n_live: Optional[torch.Tensor] = None, stats=None) -> torch.Tensor:
    """``relu(bn(features))`` (relu optional) with torch.nn.BatchNorm1d's bookkeeping.  n_live: device
    int32 scalar of a static-shape tensor (spconv_amd/pytorch/static.py) -- statistics over the first
    n_live rows, the padding rows come out as zeros in both directions.  stats: an ``ops.StatsSink`` the convolution
    that produced `features` filled from its epilogue (per-workgroup {rows, mean, M2} records of exactly these rows);
    the statistics pass over the rows is then skipped."""
    momentum = 0.0 if bn.momentum is None else bn.momentum
    nbt = None
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        if bn.momentum is None:       # cumulative average: the factor is 1 / count, needed on the host
            bn.num_batches_tracked.add_(1)
            momentum = 1.0 / float(bn.num_batches_tracked)
        elif bn.num_batches_tracked.dtype == torch.int64 and bn.num_batches_tracked.is_cuda:
            nbt = bn.num_batches_tracked          # incremented by the statistics kernel
        else:
            bn.num_batches_tracked.add_(1)
    use_batch = bn.training or (bn.running_mean is None and bn.running_var is None)
    update = bn.training and bn.track_running_stats
    differentiable = torch.is_grad_enabled() and any(t is not None and t.requires_grad
                                                     for t in (features, bn.weight, bn.bias))
    records, n_records = None, 0
    if (stats is not None and stats.records is not None and use_batch and stats.rows == features.shape[0]
            and stats.channels == features.shape[1] and stats.n_live is n_live):
        records, n_records = stats.records, stats.count
    return _BatchNormFn.apply(features, bn.weight, bn.bias, bn.running_mean if (update or not use_batch) else None,
                              bn.running_var if (update or not use_batch) else None, use_batch, momentum, bn.eps,
                              relu, nbt, n_live, differentiable, records, n_records)
