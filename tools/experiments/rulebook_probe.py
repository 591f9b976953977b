#!/usr/bin/env python
"""Builds SubM rulebooks (uniform cfg-2 scene and the LiDAR fixture) and a tile plan in a loop, for
rocprofv3 --kernel-trace --stats:  per-kernel time of the builders."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from spconv_amd.pytorch import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for kind in ("uniform", "fixture"):
        idx, shape = bench.make_scene(kind, 100_000, 0)
        ind = torch.from_numpy(idx).to(dev)
        for it in range(12):
            ops.build_rulebook(ind, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True, do_sort="layout")
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
