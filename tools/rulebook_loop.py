#!/usr/bin/env python
"""Eager loop of static-shape SparseConv k3 s2 rulebook builds for rocprofv3 --kernel-trace --stats
(per-kernel device time of the builder passes).   RB_SCENE=uniform|lidar RB_BATCH=1|4 SPX_CONV_V=2|3"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from spconv_amd.pytorch import ops  # noqa: E402
from spconv_amd.utils import nets  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    kind, bs = os.environ.get("RB_SCENE", "lidar"), int(os.environ.get("RB_BATCH", "4"))
    idx, shape = bench.make_scene(kind, 100_000, 0, batch=bs, shape=nets.SECOND_SHAPE if bs > 1 else None)
    ind = torch.from_numpy(idx).to(dev)
    rb, _ = ops.build_rulebook(ind, bs, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, [0] * 3, False)
    cap = rb.n_out + 1024
    for _ in range(20):
        ops.build_rulebook(ind, bs, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, [0] * 3, False,
                           need_native=os.environ.get("RB_NATIVE", "0") == "1" and False, static_num_out=cap)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
