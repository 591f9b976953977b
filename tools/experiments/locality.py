#!/usr/bin/env python
"""Effect of the voxel ROW ORDER on the gather kernels (L2 reuse of neighbour rows): the same
coordinate set in shuffled, voxeliser (first-seen) and sorted order.

    python tools/locality.py          # synthetic LiDAR-density scene and the reference's real scene"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import event_time_ms  # noqa: E402
from spconv_amd.pytorch import ops  # noqa: E402
from spconv_amd.utils import synthetic  # noqa: E402


def times(idx_np, shape, dev, C=64):
    idx = torch.from_numpy(np.ascontiguousarray(idx_np)).to(dev)
    n = idx.shape[0]
    f = (torch.rand(n, C, device=dev) * 2 - 1).half()
    d = ((torch.rand(n, C, device=dev) * 2 - 1) * 0.2).half()
    w = (torch.rand(C, 3, 3, 3, C, device=dev) * 2 - 1).half()
    rb, _ = ops.build_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)
    plan = ops._plan_of(rb)
    t_f = event_time_ms(lambda: ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, n, 13))
    t_b = event_time_ms(lambda: ops.igemm_bwd(f, d, w, rb.pair_fwd, rb.mask_fwd, None, rb.pair_native,
                                              rb.num_per_loc, True, plan))
    t_r = event_time_ms(lambda: ops.build_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3,
                                                   [0] * 3, True), iters=5, warm=2)
    return dict(fwd_us=round(t_f * 1e3, 1), bwd_us=round(t_b * 1e3, 1), rulebook_us=round(t_r * 1e3, 1))


def orders(idx, shape, seed=0):
    rng = np.random.default_rng(seed)
    lin = np.ravel_multi_index((idx[:, 1], idx[:, 2], idx[:, 3]), shape)
    yx_z = np.lexsort((idx[:, 1], idx[:, 3], idx[:, 2]))        # y, then x, then z (a BEV sweep)
    return {"as_given": idx, "shuffled": idx[rng.permutation(idx.shape[0])], "sorted_zyx": idx[np.argsort(lin)],
            "sorted_yxz": idx[yx_z]}


def main():
    dev = torch.device("cuda:0")
    out = {}
    shape = [40, 1280, 1600]
    syn = synthetic.lidar_like_scene(shape, 100_000, 1, seed=0)
    out["synthetic_lidar_100k"] = {k: times(v, shape, dev) for k, v in orders(syn, shape).items()}
    from golden import lidar_scene
    real, rshape = lidar_scene()
    out["reference_fixture_125k"] = {k: times(v, list(rshape), dev) for k, v in orders(real, list(rshape)).items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
