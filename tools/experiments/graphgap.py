#!/usr/bin/env python
"""How much of a hipGraph-replayed step is the gap between replays?  Captures U steps of the
bench workload per graph and reports the wall time per step for U = 1, 2, 4, 8."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spconv_amd.pytorch as spconv
from spconv_amd.pytorch import ops
from spconv_amd.utils import synthetic

SHAPE = [40, 1280, 1600]
dev = torch.device("cuda:0")
idx = torch.from_numpy(synthetic.uniform_scene(SHAPE, 100_000, 1, seed=0)).to(dev)
n = idx.shape[0]
feats = (torch.rand((n, 64)) * 2 - 1).to(dev, torch.float16).requires_grad_(True)
dout = ((torch.rand((n, 64)) * 2 - 1) * 0.2).to(dev, torch.float16)
net = spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="b").to(dev, torch.float16).train()
rb = ops.build_rulebook(idx, 1, SHAPE, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)[0]
x = spconv.SparseConvTensor(feats, idx, SHAPE, 1)
x.indice_dict["b"] = net._make_indice_data(rb, idx, SHAPE, SHAPE, net.algo)

def compute():
    net.weight.grad = None
    feats.grad = None
    net(x).features.backward(dout)

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        compute()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
res = {}
for U in (1, 2, 4, 8):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(U):
            compute()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    reps = 400 // U
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    res[U] = round((time.perf_counter() - t0) / (reps * U) * 1e6, 2)
print(json.dumps(res))
