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
