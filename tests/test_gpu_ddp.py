"""GPU: the way users of the reference actually distribute -- torch.nn.parallel.DistributedDataParallel around a voxel
backbone, optionally with SparseSyncBatchNorm (spconv/pytorch/modules.py:162-168; the reference's only distributed artefact
is test/fake_dist_train.py:113-129, which wraps its net in DDP the same way).  The custom autograd Functions, the fused
BatchNorm and the rulebook cache have to live under DDP's hooks and bucketing.

A one-GPU box cannot host two RCCL ranks, so two ranks share cuda:0 over gloo: DDP, its reducer, the bucket all-reduce
and (second test) the statistics exchange of SyncBatchNorm are the real ones, only the transport differs."""
import copy
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scenes(bs, shape, per_scene, C, seed):
    rng = np.random.default_rng(seed)
    rows = []
    for b in range(bs):
        lin = rng.choice(int(np.prod(shape)), per_scene, replace=False)
        rows.append(np.concatenate([np.full((per_scene, 1), b), np.stack(np.unravel_index(lin, shape), 1)], 1))
    idx = np.concatenate(rows).astype(np.int32)
    return idx, rng.uniform(-1, 1, (idx.shape[0], C)).astype(np.float32)


class _Backbone(torch.nn.Module):
    """The config-4 network (spconv_amd.utils.nets.second_backbone: 12 sparse convolutions + BatchNorm1d + ReLU) behind
    a tensors-in / tensor-out forward, the shape DDP's forward hooks expect."""

    def __init__(self, cin, shape, sync_bn):
        super().__init__()
        from spconv_amd.utils import nets
        torch.manual_seed(7)
        self.net = nets.second_backbone(cin)
        if sync_bn:
            self.net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(self.net)
        self.shape = shape

    def forward(self, feat, idx, batch):
        import spconv_amd.pytorch as spconv
        y = self.net(spconv.SparseConvTensor(feat, idx, self.shape, batch))
        # a fixed function of the OUTPUT coordinates weights the loss, so that sharding does not change it
        co = y.indices[:, 1:].float()
        g = torch.sin(co.sum(1, keepdim=True) * 0.37 + torch.arange(y.features.shape[1], device=feat.device).float() * 0.11)
        return (y.features.float() * g).sum()


def _grads_of(model, feat, idx, batch):
    model.zero_grad(set_to_none=True)
    model(feat, idx, batch).backward()
    return [p.grad.detach().clone() for p in model.parameters()]


def _worker(rank, world, port, sync_bn, out):
    from spconv_amd.dist import shard_scenes
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    bs, shape, C = 4, [41, 64, 64], 4
    idx_np, feat_np = _scenes(bs, shape, 6000, C, seed=3)                  # the same data on every rank
    idx, feat = torch.from_numpy(idx_np).to(dev), torch.from_numpy(feat_np).to(dev)
    base = _Backbone(C, shape, sync_bn).to(dev)                           # fp32: the comparison is about wiring, not rounding
    shards = [shard_scenes(idx, feat, bs, r, world) for r in range(world)]
    if sync_bn:
        # statistics over ALL ranks' rows: every rank's gradient is its share of the full-batch gradient with full-batch
        # statistics, and DDP's average is that gradient / world
        ref_model = copy.deepcopy(base)
        ref_model.net = _plain_bn(ref_model.net)
        want = [g / world for g in _grads_of(ref_model, feat, idx, bs)]
    else:
        # plain BatchNorm normalises per rank: DDP's result is the mean over the ranks of the per-shard gradients
        per = []
        for li, lf, lb in shards:
            m = copy.deepcopy(base)
            per.append(_grads_of(m, lf, li, lb))
        want = [torch.stack(gs).mean(0) for gs in zip(*per)]
    ddp = torch.nn.parallel.DistributedDataParallel(copy.deepcopy(base))
    li, lf, lb = shards[rank]
    got = _grads_of(ddp, lf, li, lb)
    errs = []
    for g, w in zip(got, want):
        errs.append(float((g - w).abs().max() / max(float(w.abs().max()), 1e-12)))
    # a second step under DDP (rulebook cache, reducer state and buffers survive an iteration)
    got2 = _grads_of(ddp, lf, li, lb)
    rep = max(float((a - b).abs().max()) for a, b in zip(got, got2)) if not sync_bn else 0.0
    out[rank] = (errs, rep)
    dist.barrier()
    dist.destroy_process_group()


def _plain_bn(net):
    """SyncBatchNorm -> BatchNorm1d with the same parameters (the single-process reference of the SyncBN run)."""
    for name, m in list(net.named_children()):
        if isinstance(m, torch.nn.SyncBatchNorm):
            bn = torch.nn.BatchNorm1d(m.num_features, eps=m.eps, momentum=m.momentum).to(m.weight.device)
            bn.load_state_dict(m.state_dict())
            setattr(net, name, bn)
        else:
            _plain_bn(m)
    return net


@pytest.mark.parametrize("sync_bn", [False, True])
def test_ddp_around_the_second_backbone_world2(cuda, sync_bn):
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), sync_bn, out), nprocs=world, join=True)
        res = dict(out)
    assert set(res) == {0, 1}
    for rank, (errs, rep) in res.items():
        # fp32 through 12 convolutions and 12 normalisation layers.  Plain BatchNorm: the same kernels on the same rows,
        # only DDP's averaging differs (measured < 1e-4).  SyncBatchNorm: torch's gathered-statistics kernels on one
        # side, this library's fused BatchNorm1d over the full batch on the other -- two ways of summing a variance whose
        # last-bit differences the twelve 1/sigma factors of the backward pass amplify (measured 5e-3 on the deepest,
        # smallest level; a wiring error is O(1))
        assert max(errs) < (2e-2 if sync_bn else 2e-4), (rank, errs)
        assert rep == 0.0, (rank, rep)          # plain BatchNorm: the same step twice is bit-identical under DDP
