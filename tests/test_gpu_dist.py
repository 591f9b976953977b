"""GPU: the data-parallel path with the HIP kernels under it (the reference's own multi-process
check is test/fake_dist_train.py:113-129).  A one-GPU box cannot host two RCCL ranks, so two ranks share cuda:0 over
gloo -- sharding, the per-rank rulebooks / convolutions / gradients and the flat-bucket all-reduce are the real ones,
only the transport differs."""
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


def _net(C, dev):
    import spconv_amd.pytorch as spconv
    torch.manual_seed(11)
    return spconv.SparseSequential(
        spconv.SubMConv3d(C, 16, 3, bias=False, indice_key="a"),
        spconv.SubMConv3d(16, 16, 3, bias=False, indice_key="a"),
        spconv.SparseConv3d(16, 32, 3, 2, 1, bias=False),
        spconv.SparseMaxPool3d(2, 2)).to(dev)


def _grads(net, idx, feat, batch, shape, dev):
    """Sum-of-weighted-outputs loss whose weights are a fixed function of the OUTPUT coordinates (sharding does not
    change them); returns the parameter gradients."""
    import spconv_amd.pytorch as spconv
    net.zero_grad(set_to_none=True)
    x = spconv.SparseConvTensor(feat.to(dev), idx.to(dev), shape, batch)
    y = net(x)
    co = y.indices[:, 1:].float()
    g = torch.sin(co.sum(1, keepdim=True) * 0.37 + torch.arange(y.features.shape[1], device=dev).float() * 0.11)
    (y.features * g).sum().backward()
    return [p.grad.detach().clone() for p in net.parameters()]


def _worker(rank, world, port, out):
    from spconv_amd.dist import GradBucket, shard_scenes
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    bs, shape, C = 4, [16, 24, 24], 8
    idx_np, feat_np = _scenes(bs, shape, 900, C, seed=5)                 # the same data on every rank
    idx, feat = torch.from_numpy(idx_np), torch.from_numpy(feat_np)
    net = _net(C, dev)
    full = _grads(net, idx, feat, bs, shape, dev)                         # the whole batch, this process
    li, lf, lb = shard_scenes(idx, feat, bs, rank, world)
    assert lb == bs // world and li.shape[0] == idx.shape[0] // world
    _grads(net, li, lf, lb, shape, dev)                                    # this rank's scenes: .grad of the shard
    bucket = GradBucket(net.parameters(), dtype=torch.float32)
    bucket.all_reduce(average=True)                                        # ONE flat all-reduce over the transport
    errs = []
    for p, f in zip(net.parameters(), full):
        want = f / world
        errs.append(float((p.grad - want).abs().max() / want.abs().max()))
    out[rank] = errs
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_chain_on_the_hip_kernels_gives_the_full_batch_gradient_world2(cuda):
    """SubM -> SubM (shared rulebook) -> stride-2 SparseConv -> max pool in fp32 on two ranks, two scenes each: the
    averaged gradients equal the full-batch gradients / world (scenes never interact: the batch index is part of every
    hash key, csrc/sparse/indices.py:108-109)."""
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        res = dict(out)
    assert set(res) == {0, 1}
    for rank, errs in res.items():
        assert all(e < 1e-5 for e in errs), (rank, errs)     # fp32: only the summation order over scenes differs
