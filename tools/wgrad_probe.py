#!/usr/bin/env python
"""What bounds wgrad on a large, dense level of a backbone?  Times the standalone wgrad, dgrad and the
fused backward at C = K in {16, 32, 64} on a LiDAR-like scene, with the real ConvAlgo.Native lists,
with every pair redirected to rows 0..127 (no memory traffic to speak of) and with each list
shuffled (no locality at all)."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spconv_amd.pytorch import ops

dev = torch.device("cuda:0")
n_target = int(os.environ.get("PROBE_N", "300000"))
idx, shape = bench.make_scene("lidar", n_target // 4, 0, batch=4, shape=[21, 800, 704])
ind = torch.from_numpy(idx).to(dev)
rb = ops.build_rulebook(ind, 4, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)[0]
n = idx.shape[0]
num = rb.num_per_loc.cpu().numpy()
pairs = int(n + 2 * num[:13].sum())
nat = rb.pair_native
fake = nat.clone()
j = torch.arange(n, device=dev, dtype=torch.int32) % 128
fake[:] = j
shuf = nat.clone()
cnt = [int(min(n, num[k] if k < 13 else (n if k == 13 else num[26 - k]))) for k in range(27)]
g = torch.Generator(device="cpu"); g.manual_seed(0)
for k in range(27):
    p = torch.randperm(cnt[k], generator=g).to(dev)
    shuf[0, k, :cnt[k]] = nat[0, k, :cnt[k]][p]
    shuf[1, k, :cnt[k]] = nat[1, k, :cnt[k]][p]
plan = ops.wgrad_plan(rb.num_per_loc, n, 27, True)
res = {"n": n, "pairs": pairs, "var": os.environ.get("SPX_WGRAD_VAR", "0")}
PMC = os.environ.get("PROBE_PMC") == "1"
for C in ((int(os.environ["PROBE_C"]),) if "PROBE_C" in os.environ else (16, 32, 64)):
    f = torch.randn(n, C, device=dev).half()
    d = torch.randn(n, C, device=dev).half()
    w = (torch.randn(C, 3, 3, 3, C, device=dev) * 0.1).half()
    t = (lambda fn: round(1e3 * bench.event_time_ms(fn, iters=6, warm=2, span=0), 1)) if PMC else (lambda fn: round(1e3 * bench.event_time_ms(fn, span=4), 1))
    r = {}
    r["wgrad"] = t(lambda i: ops.igemm_wgrad(f, d, w.shape, nat, rb.num_per_loc, True, plan))
    if os.environ.get("PROBE_ONLY_WGRAD") == "1":
        res[f"C{C}"] = r["wgrad"]
        continue
    r["wgrad_rows0_127"] = t(lambda i: ops.igemm_wgrad(f, d, w.shape, fake, rb.num_per_loc, True, plan))
    r["wgrad_shuffled"] = t(lambda i: ops.igemm_wgrad(f, d, w.shape, shuf, rb.num_per_loc, True, plan))
    r["dgrad"] = t(lambda i: ops.igemm_dgrad(d, w, rb.pair_fwd, rb.mask_fwd, None, n, True))
    r["fused_bwd"] = t(lambda i: ops.igemm_bwd(f, d, w, rb.pair_fwd, rb.mask_fwd, None, nat, rb.num_per_loc, True, plan))
    r["wgrad_ps_per_pair"] = round(r["wgrad"] * 1e6 / pairs, 1)
    res[f"C{C}"] = r
print(json.dumps(res))
