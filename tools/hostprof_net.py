#!/usr/bin/env python
"""cProfile of the eager training step of bench.py's network configurations (3: stride-2 chain on the
fixture, 4: SECOND-style backbone): where the host time of a step goes.
    python tools/hostprof_net.py 3|4"""
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
from spconv_amd.utils import nets  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
dev = torch.device("cuda:0")
torch.manual_seed(0)
if cfg == "3":
    net = nets.downsample_chain().to(dev).half().train()
    idx_np, shape = bench.make_scene("fixture", 100_000, seed=0)
    cin, bs = 16, 1
else:
    net = nets.second_backbone(4).to(dev).half().train()
    idx_np, shape = bench.make_scene("lidar", 100_000, seed=0, batch=4, shape=nets.SECOND_SHAPE)
    cin, bs = 4, 4
ind = torch.from_numpy(idx_np).to(dev)
f = torch.randn(idx_np.shape[0], cin, device=dev).half()
g = {}


def step():
    net.zero_grad(set_to_none=True)
    x = spconv.SparseConvTensor(f.clone().requires_grad_(cfg == "3"), ind, shape, bs)
    y = net(x)
    go = g.get(y.features.shape)
    if go is None:
        go = g[y.features.shape] = (torch.rand(y.features.shape, device=dev) - 0.5).half() * 0.2
    y.features.backward(go)


for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
print("config %s: %.3f ms per step host-side, %.3f ms drained" % (cfg, th / 50 * 1e3, (time.perf_counter() - t0) / 50 * 1e3), flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    step()
pr.disable()
torch.cuda.synchronize()
sio = io.StringIO()
pstats.Stats(pr, stream=sio).sort_stats("tottime").print_stats(40)
print("---- cProfile of 30 steps (tottime; ms per step = tottime / 30 * 1000)")
print("\n".join(l[:150] for l in sio.getvalue().splitlines()[4:52]), flush=True)
