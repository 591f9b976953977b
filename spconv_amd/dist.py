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
    """Flat gradient bucket: one all-reduce for every parameter of a module."""

    def __init__(self, params: Iterable[torch.nn.Parameter], dtype: torch.dtype = torch.float32):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(self.numel, dtype=dtype, device=dev)

    def all_reduce(self, group: Optional[dist.ProcessGroup] = None, average: bool = True):
        """Packs .grad of every parameter, all-reduces once, unpacks in place."""
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            if average:
                self.flat.div_(dist.get_world_size(group))
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(self.flat[off:off + n].view_as(p))
            off += n
        return self.flat
