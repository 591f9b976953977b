"""CPU, world_size 2 over gloo: scene sharding and the single-bucket gradient all-reduce."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spconv_amd.dist import GradBucket, scenes_for_rank, shard_scenes


def test_scene_partition_is_a_partition():
    for bs in (1, 7, 8, 32):
        for world in (1, 2, 3, 8):
            got = [s for r in range(world) for s in scenes_for_rank(bs, r, world)]
            assert got == list(range(bs))


def test_shard_scenes_rebases_batch_ids():
    rng = np.random.default_rng(0)
    idx = torch.from_numpy(np.concatenate(
        [np.concatenate([np.full((5, 1), b), rng.integers(0, 9, (5, 3))], 1) for b in range(6)]).astype(np.int32))
    feat = torch.arange(30, dtype=torch.float32).view(30, 1)
    seen = []
    for r in range(4):
        i, f, lb = shard_scenes(idx, feat, 6, r, 4)
        own = scenes_for_rank(6, r, 4)
        assert lb == len(own) and i.shape[0] == 5 * lb
        assert (i[:, 0] >= 0).all() and (i[:, 0] < max(lb, 1)).all()
        seen.append(f)
    assert torch.equal(torch.cat(seen).view(-1), feat.view(-1))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)                                   # same init on every rank
    net = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.Linear(8, 2))
    x = torch.full((3, 4), float(rank + 1))
    net(x).sum().backward()
    local = [p.grad.clone() for p in net.parameters()]
    bucket = GradBucket(net.parameters())
    bucket.all_reduce(average=True)
    gathered = [torch.zeros(bucket.numel) for _ in range(world)]
    dist.all_gather(gathered, torch.cat([g.reshape(-1) for g in local]))
    expect = torch.stack(gathered).mean(0)
    got = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    ok = bool(torch.allclose(got, expect, atol=1e-6))
    # single parameter: reduced in place, no flat buffer (the benchmark layer's path)
    lin = torch.nn.Linear(4, 3, bias=False)
    lin(x).sum().backward()
    mine = lin.weight.grad.clone()
    b1 = GradBucket(lin.parameters())
    b1.all_reduce(average=True)
    g1 = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(g1, mine)
    ok = ok and b1.flat is None and bool(torch.allclose(lin.weight.grad, torch.stack(g1).mean(0), atol=1e-6))
    out[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_grad_bucket_all_reduce_gloo_world2():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def _conv_worker(rank, world, port, out):
    """A 2-layer conv-only chain (SubM 3x3x3 -> stride-2 SparseConv) on this rank's scenes: the
    averaged dW of the sharded run must equal the full-batch dW / world (scenes never interact)."""
    import oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(7)                         # same data on every rank
    bs, shape, C, K1, K2 = 4, [10, 12, 14], 4, 6, 8
    rows = []
    for b in range(bs):
        lin = rng.choice(int(np.prod(shape)), 150, replace=False)
        zyx = np.stack(np.unravel_index(lin, shape), 1)
        rows.append(np.concatenate([np.full((150, 1), b), zyx], 1))
    idx = np.concatenate(rows).astype(np.int32)
    feat = rng.uniform(-1, 1, (idx.shape[0], C)).astype(np.float32)
    w1 = torch.from_numpy(rng.uniform(-1, 1, (K1, 3, 3, 3, C)).astype(np.float32))
    w2 = torch.from_numpy(rng.uniform(-1, 1, (K2, 3, 3, 3, K1)).astype(np.float32))

    def chain_grads(idx_np, feat_np, batch):
        """forward through both layers, loss = sum(out * g) with g a fixed function of the OUTPUT
        coordinates (so that sharding does not change it); returns (dW1, dW2)."""
        f = torch.from_numpy(feat_np)
        o1_inds, p1, n1, _ = oracle.get_indice_pairs(idx_np, batch, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, subm=True)
        y1 = oracle.indice_conv(f, w1, p1, n1, idx_np.shape[0], subm=True)
        o2_inds, p2, n2, oshape = oracle.get_indice_pairs(idx_np, batch, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3,
                                                          subm=False)
        y2 = oracle.indice_conv(y1, w2, p2, n2, o2_inds.shape[0], subm=False)
        co = torch.from_numpy(o2_inds[:, 1:].astype(np.float32))
        g = torch.sin(co.sum(1, keepdim=True) * 0.37 + torch.arange(K2).float() * 0.11)    # [n_out, K2]
        assert g.shape == y2.shape
        d1, dw2 = oracle.indice_conv_backward(y1, w2, g, p2, n2, subm=False)
        _, dw1 = oracle.indice_conv_backward(f, w1, d1, p1, n1, subm=True)
        return dw1, dw2
    full1, full2 = chain_grads(idx, feat, bs)
    li, lf, lb = shard_scenes(torch.from_numpy(idx), torch.from_numpy(feat), bs, rank, world)
    m1, m2 = chain_grads(li.numpy(), lf.numpy(), lb)
    p1, p2 = torch.nn.Parameter(w1.clone()), torch.nn.Parameter(w2.clone())
    p1.grad, p2.grad = m1.clone(), m2.clone()
    GradBucket([p1, p2]).all_reduce(average=True)
    ok = bool(torch.allclose(p1.grad, full1 / world, rtol=1e-4, atol=1e-5)
              and torch.allclose(p2.grad, full2 / world, rtol=1e-4, atol=1e-5))
    out[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_conv_chain_gives_the_full_batch_gradient_gloo_world2():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_conv_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}
