#!/usr/bin/env python
"""Host-side cost of one single-scene backbone training step, split into forward (autograd on),
loss and backward, with a cumulative cProfile of the forward and torch's own profiler for the
backward thread.  Companion of tools/experiments/hostprof.py."""
import cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spconv_amd.pytorch as spconv
from spconv_amd.utils import nets, synthetic

dev = torch.device("cuda:0")
idx = torch.from_numpy(synthetic.lidar_like_scene(nets.SECOND_SHAPE, 100_000, 1, seed=0)).to(dev)
n = idx.shape[0]
net = nets.second_backbone(4).to(dev).half().train()
f4 = torch.randn(n, 4, device=dev).half()
g = {}

def fwd():
    net.zero_grad(set_to_none=True)
    return net(spconv.SparseConvTensor(f4, idx, nets.SECOND_SHAPE, 1))

def bwd(out):
    go = g.get(out.features.shape)
    if go is None:
        go = g[out.features.shape] = torch.randn_like(out.features) * 0.1
    out.features.backward(go)

def timeit(fn, iters=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    th = (time.perf_counter() - t0) / iters * 1e3
    torch.cuda.synchronize()
    return th, (time.perf_counter() - t0) / iters * 1e3

def nograd():
    with torch.no_grad():
        fwd()
print("forward, no_grad (train mode): host %.2f ms, drained %.2f ms" % timeit(nograd))
print("forward, autograd on:          host %.2f ms, drained %.2f ms" % timeit(fwd))
print("forward + backward:            host %.2f ms, drained %.2f ms" % timeit(lambda: bwd(fwd())))

def synced_step():
    out = fwd(); torch.cuda.synchronize(); t0 = time.perf_counter()
    bwd(out); t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) * 1e3
for _ in range(3): synced_step()
print("backward alone (GPU idle at start): host %.2f ms" % (sum(synced_step() for _ in range(10)) / 10))

pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    out = fwd()
pr.disable(); torch.cuda.synchronize()
sio = io.StringIO(); pstats.Stats(pr, stream=sio).sort_stats("tottime").print_stats(45)
print("\n".join(l[:160] for l in sio.getvalue().splitlines()[4:56]))

from torch.profiler import profile, ProfilerActivity
out = fwd()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(5):
        bwd(fwd())
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60))
