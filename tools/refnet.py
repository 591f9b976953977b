#!/usr/bin/env python
"""The reference's own benchmark network, timed the reference's way, beside its published numbers.

`python -m spconv.benchmark bench_basic f16` (spconv/benchmark/basic.py:16-194): 14 x SubMConv3d 3x3x3
(3 -> 64 -> 64 | 96, 96 | 128, 128 | 160, 160 | 192, 192 | 224, 224 | 256, 256) with 6 x SparseMaxPool3d(2, 2)
between the stages, no bias, no normalisation, fp16, on the real-LiDAR fixture test/data/test_spconv.pkl
("Basic (120k voxels)": 125 562 voxels -- its coordinates are tests/golden/lidar_scene.npz; the 3-channel
voxel features are not committed, random values stand in, which changes no kernel's work).
Protocol as basic.py:166-194: forward = mean of the last 50 of 100 no-grad iterations, backward = mean of
the last 25 of 50 `out.features.backward(dout)` calls, wall clock with a device synchronisation on both
sides (tv.measure_duration).  docs/BENCHMARK.md:25-32 publishes forward / backward [ms]:
A100 13.02 / 12.43, RTX 4090 7.37 / 6.87, RTX 3090 11.84 / 11.84, V100-32G 15.55 / 14.90.
Context, not a same-node race: those are other GPUs running spconv's CUDA kernels.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import spconv_amd.pytorch as spconv  # noqa: E402

PUBLISHED_MS = {"A100": (13.02, 12.43), "RTX 4090": (7.37, 6.87), "RTX 3090": (11.84, 11.84),
                "V100-32G": (15.55, 14.90), "T4": (18.74, 25.51)}


def net(algo):
    widths = [64, 96, 128, 160, 192, 224, 256]
    layers, cin = [], 3
    for i, c in enumerate(widths):
        layers += [spconv.SubMConv3d(cin, c, 3, bias=False, indice_key=f"c{i}", algo=algo),
                   spconv.SubMConv3d(c, c, 3, bias=False, indice_key=f"c{i}", algo=algo)]
        cin = c
        if i < len(widths) - 1:
            layers.append(spconv.SparseMaxPool3d(2, 2, algo=algo))
    return spconv.SparseSequential(*layers)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3


def main():
    dev = torch.device("cuda:0")
    idx, shape = bench.fixture_scene(0)
    coors = torch.from_numpy(idx).to(dev)
    rng = np.random.default_rng(0)
    voxels = torch.from_numpy(rng.uniform(-1, 1, (idx.shape[0], 3)).astype(np.float32)).to(dev).half()
    voxels.requires_grad = True
    res = {"voxels": int(idx.shape[0]), "dtype": "f16", "published_ms_fwd_bwd": PUBLISHED_MS}
    for algo in (spconv.ConvAlgo.Native, spconv.ConvAlgo.MaskImplicitGemm):
        torch.manual_seed(0)
        m = net(algo).to(dev).train().half()
        with torch.no_grad():
            out = m(spconv.SparseConvTensor(voxels, coors, shape, 1))
        dout = torch.from_numpy(rng.uniform(-0.2, 0.2, tuple(out.features.shape)).astype(np.float32)).to(dev).half()
        fwd = []
        with torch.no_grad():
            for _ in range(100):
                fwd.append(timed(lambda: m(spconv.SparseConvTensor(voxels, coors, shape, 1)))[1])
        bwd = []
        for _ in range(50):
            o = m(spconv.SparseConvTensor(voxels, coors, shape, 1))
            bwd.append(timed(lambda: o.features.backward(dout))[1])
        res[str(algo).split(".")[-1]] = {"forward_ms": round(float(np.mean(fwd[50:])), 3),
                                         "backward_ms": round(float(np.mean(bwd[25:])), 3),
                                         "out_voxels": int(out.features.shape[0])}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
