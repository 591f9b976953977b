#!/usr/bin/env python
"""Fused-backward time on one scene for the current SPX_* environment (A/B runs of the wgrad
group count / fusion switch): python tools/gsweep.py [fixture|lidar|uniform] [voxels]"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import event_time_ms
from spconv_amd.pytorch import ops
from spconv_amd.utils import synthetic
kind = sys.argv[1] if len(sys.argv) > 1 else "fixture"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
dev = torch.device("cuda:0")
if kind == "fixture":
    from golden import lidar_scene
    idx_np, shape = lidar_scene(); shape = list(shape)
else:
    shape = [40, 1280, 1600]
    idx_np = (synthetic.lidar_like_scene if kind == "lidar" else synthetic.uniform_scene)(shape, n, 1, seed=0)
idx = torch.from_numpy(np.ascontiguousarray(idx_np)).to(dev); n = idx.shape[0]
C = int(os.environ.get("GS_CHANNELS", "64"))
f = (torch.rand(n, C, device=dev) * 2 - 1).half(); d = ((torch.rand(n, C, device=dev) * 2 - 1) * 0.2).half()
w = (torch.rand(C, 3, 3, 3, C, device=dev) * 2 - 1).half()
rb, _ = ops.build_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)
plan = ops._plan_of(rb)
t_b = event_time_ms(lambda: ops.igemm_bwd(f, d, w, rb.pair_fwd, rb.mask_fwd, None, rb.pair_native, rb.num_per_loc, True, plan))
t_d = event_time_ms(lambda: ops.igemm_dgrad(d, w, rb.pair_fwd, rb.mask_fwd, None, n, True))
t_w = event_time_ms(lambda: ops.igemm_wgrad(f, d, w.shape, rb.pair_native, rb.num_per_loc, True, plan))
print(json.dumps({"scene": kind, "n": n, "env": {k: v for k, v in os.environ.items() if k.startswith("SPX_")},
                  "bwd_us": round(t_b * 1e3, 1), "dgrad_us": round(t_d * 1e3, 1), "wgrad_us": round(t_w * 1e3, 1)}))
