#!/usr/bin/env python
"""What the ROW ORDER of a level costs: the levels behind a strided layer of the config-4 backbone (4 LiDAR-like
scenes) in the order the strided builder numbers them (first seen, i.e. the shuffled order of its input) against the
same voxels sorted by their linear coordinate key (the order of the reference's sort + unique path,
spconv/csrc/sparse/all.py:1533-1552).  SubM forward / dgrad / fused backward per level, device time over graph replays.
    python tools/order_probe.py      -> one JSON line per (level, order)"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spconv_amd.pytorch import ops
from spconv_amd.utils import nets

dev = torch.device("cuda:0")
idx, shape = bench.make_scene("lidar", 100_000, 0, batch=4, shape=nets.SECOND_SHAPE)
ind = torch.from_numpy(idx).to(dev)
t = lambda fn: round(1e3 * bench.event_time_ms(fn, span=4), 1)


def sorted_rows(ind, shape):
    key = ind[:, 0].long()
    for d, s in enumerate(shape):
        key = key * s + ind[:, 1 + d].long()
    return ind[torch.argsort(key)].contiguous()


def probe(level, C, ind, shape, order):
    n = ind.shape[0]
    rb = ops.build_rulebook(ind, 4, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)[0]
    f = torch.randn(n, C, device=dev).half()
    d = torch.randn(n, C, device=dev).half()
    w = (torch.randn(C, 3, 3, 3, C, device=dev) * 0.1).half()
    plan = ops._plan_of(rb)
    r = dict(level=level, C=C, order=order, voxels=n)
    r["rulebook_us"] = t(lambda i: ops.build_rulebook(ind, 4, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True))
    r["fwd_us"] = t(lambda i: ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, n, 13))
    r["dgrad_us"] = t(lambda i: ops.igemm_dgrad(d, w, rb.pair_fwd, rb.mask_fwd, None, n, True))
    r["bwd_us"] = t(lambda i: ops.igemm_bwd(f, d, w, rb.pair_fwd, rb.mask_fwd, None, rb.pair_native, rb.num_per_loc, True, plan))
    # the strided layer behind this level (gathers THIS level's rows, numbers the next level)
    rb2, shape2 = ops.build_rulebook(ind, 4, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, [0] * 3, False)
    n2 = rb2.out_indices.shape[0]
    w2 = (torch.randn(2 * C, 3, 3, 3, C, device=dev) * 0.1).half()
    r["down_fwd_us"] = t(lambda i: ops.igemm_fwd(f, w2, rb2.pair_fwd, rb2.mask_fwd, None, n2, -1))
    print(json.dumps(r), flush=True)
    return rb2.out_indices, shape2


# level 1 as the user hands it over (shuffled) and sorted, for reference
probe(1, 16, ind, shape, "shuffled (as given)")
probe(1, 16, sorted_rows(ind, shape), shape, "sorted")
ind2, shape2 = ops.build_rulebook(ind, 4, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, [0] * 3, False)[0].out_indices, None
rb12, shape2 = ops.build_rulebook(ind, 4, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, [0] * 3, False)
ind2 = rb12.out_indices
ind3, shape3 = probe(2, 32, ind2, shape2, "first seen")
ind3s, _ = probe(2, 32, sorted_rows(ind2, shape2), shape2, "sorted")
probe(3, 64, ind3, shape3, "first seen")
probe(3, 64, sorted_rows(ind3, shape3), shape3, "sorted (of first-seen level 3)")
probe(3, 64, sorted_rows(ind3s, shape3), shape3, "sorted")
