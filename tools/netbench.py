#!/usr/bin/env python
"""Whole-network timings for BASELINE configs 3 and 4 (parity lives in tests/; this is the
"how long does a real backbone step take" number quoted in DESIGN.md).

    python tools/netbench.py [uniform|lidar] [batch]

cfg 3: SparseConv3d k3 s2 p1 chain 16 -> 32 -> 64 -> 128 (fp16, forward only + rulebooks).
cfg 4: SECOND-style VoxelBackBone8x (SubM/SparseConv stack 16-32-64-64-128 with BatchNorm1d +
ReLU), fp16 autocast-free half model, forward + backward, fresh rulebooks
every iteration (a new scene per step, as in training)."""
import json
import os
import sys
import time

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spconv_amd.pytorch as spconv  # noqa: E402
from spconv_amd.utils import synthetic  # noqa: E402

SHAPE = [41, 1600, 1408]


def block(cin, cout, key, n=2):
    layers = []
    for i in range(n):
        layers += [spconv.SubMConv3d(cin if i == 0 else cout, cout, 3, bias=False, indice_key=key),
                   nn.BatchNorm1d(cout, eps=1e-3, momentum=0.01), nn.ReLU()]
    return layers


def down(cin, cout, key, k=3, s=2, p=1):
    return [spconv.SparseConv3d(cin, cout, k, s, p, bias=False, indice_key=key),
            nn.BatchNorm1d(cout, eps=1e-3, momentum=0.01), nn.ReLU()]


def backbone(cin):
    return spconv.SparseSequential(
        *block(cin, 16, "subm1", 1), *block(16, 16, "subm1", 1),
        *down(16, 32, "spconv2"), *block(32, 32, "subm2"),
        *down(32, 64, "spconv3"), *block(64, 64, "subm3"),
        *down(64, 64, "spconv4", 3, 2, (0, 1, 1)), *block(64, 64, "subm4"),
        *down(64, 128, "spconv_down2", (3, 1, 1), (2, 1, 1), 0))


def timed(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "lidar"
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda:0")
    gen = synthetic.uniform_scene if scene == "uniform" else synthetic.lidar_like_scene
    idx = torch.from_numpy(gen(SHAPE, 100_000, bs, seed=0)).to(dev)
    n = idx.shape[0]
    res = {"scene": scene, "batch": bs, "voxels": n}
    # cfg 3
    torch.manual_seed(0)
    chain = spconv.SparseSequential(*[spconv.SparseConv3d(a, b, 3, 2, 1, bias=False)
                                      for a, b in ((16, 32), (32, 64), (64, 128))]).to(dev).half().eval()
    f16 = torch.randn(n, 16, device=dev).half()

    def run_chain():
        with torch.no_grad():
            return chain(spconv.SparseConvTensor(f16, idx, SHAPE, bs))
    y = run_chain()
    res["cfg3"] = {"ms_fwd_with_rulebooks": round(timed(run_chain), 3), "out_voxels": y.features.shape[0]}
    # cfg 4
    net = backbone(4).to(dev).half()
    f4 = torch.randn(n, 4, device=dev).half()

    def step():
        net.zero_grad(set_to_none=True)
        out = net(spconv.SparseConvTensor(f4, idx, SHAPE, bs))
        out.features.float().square().mean().backward()
        return out
    out = step()
    ms = timed(step)
    res["cfg4"] = {"ms_fwd_bwd_with_rulebooks": round(ms, 3), "out_voxels": out.features.shape[0],
                   "voxels_per_s": round(n / ms * 1e3)}
    net.eval()

    def infer():
        with torch.no_grad():
            return net(spconv.SparseConvTensor(f4, idx, SHAPE, bs))
    res["cfg4"]["ms_inference"] = round(timed(infer), 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
