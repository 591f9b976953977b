#!/usr/bin/env python
"""Diagnostic: per-parameter gradient of a padded (static-shape) training step against the eager step."""
import copy, os, sys
import numpy as np
import torch
from torch import nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import spconv_amd.pytorch as spconv
from test_gpu_static import _backbone, _scene_tensors

dev = torch.device("cuda:0")
shape, bs, C = [32, 40, 40], 2, 8
pad = int(os.environ.get("PAD", "1234"))
net = _backbone(spconv, C, dev, torch.float16).train()
if os.environ.get("NOSTRIDE") == "1":
    net = spconv.SparseSequential(*list(net.children())[:6]).train()
eager = copy.deepcopy(net)
f, idx = _scene_tensors(shape, 4000, bs, C, 5, dev, torch.float16)
n = f.shape[0]
fe = f.clone().requires_grad_(True)
ye = eager(spconv.SparseConvTensor(fe, idx, shape, bs))
n_out = ye.features.shape[0]
from spconv_amd.pytorch.static import strided_layers
for m in strided_layers(net).values():
    m.static_num_out = 13_000
K = ye.features.shape[1]
g = ((torch.rand((13_000 if strided_layers(net) else n + pad, K), device=dev) - 0.5) * 0.2).half()
ye.features.backward(g[:n_out])
fs = torch.zeros((n + pad, C), dtype=torch.float16, device=dev)
fs[:n] = f
fs.requires_grad_(True)
ids = torch.full((n + pad, 4), -1, dtype=torch.int32, device=dev)
ids[:n] = idx
x = spconv.SparseConvTensor(fs, ids, shape, bs)
x.n_live_dev = torch.tensor([n], dtype=torch.int32, device=dev)
ys = net(x)
ys.features.backward(g[:ys.features.shape[0]])
print("n", n, "n_out", n_out, "static rows", ys.features.shape[0])
print("out rel", float((ys.features[:n_out].float() - ye.features.float()).norm() / ye.features.float().norm()))
for (name, pa), pb in zip(net.named_parameters(), eager.parameters()):
    a, b = pa.grad.float().flatten(), pb.grad.float().flatten()
    print(f"{name:12s} rel {float((a - b).norm() / b.norm()):.5f} scale {float((a @ b) / (b @ b)):.5f}")
a, b = fs.grad[:n].float().flatten(), fe.grad.float().flatten()
print(f"din rel {float((a - b).norm() / b.norm()):.5f} scale {float((a @ b) / (b @ b)):.5f}")
# noise floor: the same eager step on the same scene with its rows permuted (mathematically identical gradients)
perm = torch.randperm(n, device=dev)
eager2 = copy.deepcopy(eager)
eager2.zero_grad(set_to_none=True)
fp = f[perm].clone().requires_grad_(True)
yp = eager2(spconv.SparseConvTensor(fp, idx[perm].contiguous(), shape, bs))
# match output rows by coordinate
def key(ind):
    return ((ind[:, 0].long() * 64 + ind[:, 1]) * 64 + ind[:, 2]) * 64 + ind[:, 3]
ke, kp = key(ye.indices), key(yp.indices)
order = torch.argsort(kp)[torch.argsort(torch.argsort(ke))]
assert torch.equal(kp[order], ke)
gp = torch.empty_like(g[:n_out])
gp[order] = g[:n_out]
yp.features.backward(gp)
print("--- eager vs eager on permuted rows")
for (name, pa), pb in zip(eager2.named_parameters(), eager.parameters()):
    a, b = pa.grad.float().flatten(), pb.grad.float().flatten()
    print(f"{name:12s} rel {float((a - b).norm() / b.norm()):.5f}")
