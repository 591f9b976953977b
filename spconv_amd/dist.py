"""Data-parallel helpers: one process per GPU, scenes sharded across ranks.

The reference has no distributed code; users wrap models in
PyTorch DDP.  The hot path shards embarrassingly: the batch index is part of
every hash key (csrc/sparse/indices.py:108-109,738), so scenes never interact
and no collective is needed inside the op.  The only exchange of a training
step is the gradient all-reduce, done here as ONE flat bucket (a SECOND-size
backbone is 3-6 MB of fp32 gradients: a single latency-bound RCCL call over
xGMI beats per-tensor calls).
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


def scenes_for_rank(batch_size: int, rank: int, world_size: int) -> range:
    """Contiguous block of scene ids owned by `rank` (first ranks get the remainder)."""
    q, r = divmod(batch_size, world_size)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def shard_scenes(indices: torch.Tensor, features: torch.Tensor, batch_size: int, rank: int,
                 world_size: int) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """Rows of the scenes owned by `rank`, with batch ids re-based to 0..local_batch-1."""
    own = scenes_for_rank(batch_size, rank, world_size)
    b = indices[:, 0]
    sel = (b >= own.start) & (b < own.stop)
    idx = indices[sel].clone()
    idx[:, 0] -= own.start
    return idx.contiguous(), features[sel].contiguous(), len(own)


class GradBucket:
    """One all-reduce for every parameter of a module.

    * a single parameter (the benchmark layer): its ``.grad`` is reduced in place with
      ``ReduceOp.AVG`` -- one RCCL call, no pack / scale / unpack kernels;
    * several parameters: gradients are packed into one flat buffer (``dtype``, default the
      first gradient's dtype), reduced once and copied back."""

    def __init__(self, params: Iterable[torch.nn.Parameter], dtype: Optional[torch.dtype] = None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.dtype = dtype
        self.flat: Optional[torch.Tensor] = None

    def _ensure_flat(self) -> torch.Tensor:
        if self.flat is None:
            dev = self.params[0].device if self.params else torch.device("cpu")
            dt = self.dtype or (self.params[0].dtype if self.params else torch.float32)
            self.flat = torch.zeros(self.numel, dtype=dt, device=dev)
        return self.flat

    @staticmethod
    def _reduce(t: torch.Tensor, group, average: bool) -> None:
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
            return
        if average and dist.get_backend(group) == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)      # RCCL averages in the kernel
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            if average:
                t.div_(dist.get_world_size(group))

    def all_reduce(self, group: Optional[dist.ProcessGroup] = None, average: bool = True):
        """All-reduces the gradients of every parameter (in place); returns the reduced buffer."""
        if len(self.params) == 1 and self.params[0].grad is not None and (
                self.dtype is None or self.dtype == self.params[0].grad.dtype):
            g = self.params[0].grad
            if not g.is_contiguous():
                g = g.contiguous()
                self.params[0].grad = g
            self._reduce(g, group, average)
            return g.view(-1)
        flat = self._ensure_flat()
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                flat[off:off + n].zero_()
            else:
                flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        self._reduce(flat, group, average)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(flat[off:off + n].view_as(p))
            off += n
        return flat
