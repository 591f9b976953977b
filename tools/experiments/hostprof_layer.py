#!/usr/bin/env python
"""Host-side cost of ONE SubMConv3d training step (BASELINE config 2, rulebook reused through
indice_key): wall time per forward / backward with the GPU drained before each, and a cProfile of
the forward call chain -- the numbers behind bench.py's `eager_device_ms_per_step`."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spconv_amd.pytorch as spconv  # noqa: E402
from spconv_amd.pytorch import ops  # noqa: E402

dev = torch.device("cuda:0")
idx_np, shape = bench.make_scene("uniform", 100_000, seed=0)
ind = torch.from_numpy(idx_np).to(dev)
net = spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="k").to(dev).half().train()
f = torch.randn(idx_np.shape[0], 64, device=dev).half().requires_grad_(True)
dout = torch.randn(idx_np.shape[0], 64, device=dev).half()
rb = ops.build_rulebook(ind, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)[0]
x = spconv.SparseConvTensor(f, ind, shape, 1)
x.indice_dict["k"] = net._make_indice_data(rb, ind, shape, shape, net.algo)
ops._plan_of(rb)


def fwd():
    return net(x)


def step():
    net.weight.grad = None
    f.grad = None
    y = fwd()
    y.features.backward(dout)


for _ in range(20):
    step()
torch.cuda.synchronize()
tf = tb = 0.0
N = 200
for _ in range(N):
    net.weight.grad = None
    f.grad = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = fwd()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    y.features.backward(dout)
    t3 = time.perf_counter()
    tf += t1 - t0
    tb += t3 - t2
print("host time per call (GPU idle at start): forward %.1f us, backward %.1f us" % (tf / N * 1e6, tb / N * 1e6))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    step()
torch.cuda.synchronize()
print("eager step, back to back: %.1f us" % ((time.perf_counter() - t0) / N * 1e6))
for name, fn in (("forward", fwd), ("step", step)):
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        fn()
    pr.disable()
    torch.cuda.synchronize()
    sio = io.StringIO()
    pstats.Stats(pr, stream=sio).sort_stats("tottime").print_stats(28)
    print(f"---- cProfile of 200 x {name} (tottime, us per call = tottime * 5000)")
    print("\n".join(l[:150] for l in sio.getvalue().splitlines()[4:40]))

