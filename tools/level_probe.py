#!/usr/bin/env python
"""Backward kernels on the REAL levels of the config-4 backbone: level 1 = 4 LiDAR-like scenes of 100 k voxels
(C = 16), level 2 = their k3 s2 p1 outputs (C = 32), level 3 = the next (C = 64).  Times forward, dgrad, wgrad
and the fused backward of a SubM 3x3x3 layer per level; pairs per voxel; achieved bytes / s against the
algorithmic bytes."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spconv_amd.pytorch import ops
from spconv_amd.utils import nets

dev = torch.device("cuda:0")
idx, shape = bench.make_scene("lidar", 100_000, 0, batch=4, shape=nets.SECOND_SHAPE)
ind = torch.from_numpy(idx).to(dev)
res = []
for level, C in ((1, 16), (2, 32), (3, 64)):
    rb = ops.build_rulebook(ind, 4, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)[0]
    n = ind.shape[0]
    num = rb.num_per_loc.cpu().numpy()
    pairs = int(n + 2 * num[:13].sum())
    f = torch.randn(n, C, device=dev).half()
    d = torch.randn(n, C, device=dev).half()
    w = (torch.randn(C, 3, 3, 3, C, device=dev) * 0.1).half()
    plan = ops._plan_of(rb)
    t = lambda fn: round(1e3 * bench.event_time_ms(fn, span=4), 1)
    r = dict(level=level, C=C, voxels=n, pairs_per_voxel=round(pairs / n, 2))
    r["fwd_us"] = t(lambda i: ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, n, 13))
    r["dgrad_us"] = t(lambda i: ops.igemm_dgrad(d, w, rb.pair_fwd, rb.mask_fwd, None, n, True))
    r["wgrad_us"] = t(lambda i: ops.igemm_wgrad(f, d, w.shape, rb.pair_native, rb.num_per_loc, True, plan))
    ops._BWD_ROWS = False
    r["fused_bwd_us"] = t(lambda i: ops.igemm_bwd(f, d, w, rb.pair_fwd, rb.mask_fwd, None, rb.pair_native, rb.num_per_loc, True, plan))
    if C <= 32:                       # the one-gather backward of narrow layers (csrc/igemm_bwdn.hip), forced on
        ops._BWD_ROWS = True
        r["bwd_rows_us"] = t(lambda i: ops.igemm_bwd(f, d, w, rb.pair_fwd, rb.mask_fwd, None, rb.pair_native, rb.num_per_loc, True, plan))
        r["occupancy"] = round(n / (4.0 * shape[0] * shape[1] * shape[2]), 5)
        r["auto_picks_rows"] = ops._dense_rows(rb, n, "fwd")
    ops._BWD_ROWS = "auto"
    ab = bench.algorithmic_bytes(n, n, C, C, 27, 2)
    r["fwd_frac"] = round(ab["fwd"] / (r["fwd_us"] * 1e-6) / 8e12, 3)
    r["bwd_frac"] = round(ab["bwd"] / (r["fused_bwd_us"] * 1e-6) / 8e12, 3)
    r["ps_per_pair_fwd"] = round(r["fwd_us"] * 1e6 / pairs, 1)
    r["ps_per_pair_bwd"] = round(r["fused_bwd_us"] * 1e6 / pairs, 1)
    res.append(r)
    # next level: k3 s2 p1
    rb2, shape = ops.build_rulebook(ind, 4, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, [0] * 3, False)
    ind = rb2.out_indices
print(json.dumps(res))
