#!/usr/bin/env python
"""Upper bound of what packing several offsets into one 128-byte reduction step could buy for
narrow layers: forward time with the full rulebook masks vs with masks thinned to 1/2 and 1/4 of
the offsets (same kernel, fewer steps)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import SHAPE, event_time_ms
from spconv_amd.pytorch import ops
from spconv_amd.utils import synthetic
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
idx = torch.from_numpy(synthetic.lidar_like_scene(SHAPE, n // 4, 4, seed=0)).to(dev)
n = idx.shape[0]
rb, _ = ops.build_rulebook(idx, 4, SHAPE, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)
out = {}
for C in (16, 32, 64):
    f = torch.randn(n, C, device=dev).half()
    w = torch.randn(C, 3, 3, 3, C, device=dev).half()
    row = {}
    for name, keep in (("all27", 0x7FFFFFF), ("half14", 0x2AAAAAA | (1 << 13)), ("quarter7", 0x0888888 | (1 << 13))):
        m = (rb.mask_fwd & keep).contiguous()
        row[name] = round(event_time_ms(lambda: ops.igemm_fwd(f, w, rb.pair_fwd, m, None, n, 13)) * 1e3, 1)
    out[f"C{C}"] = row
print(json.dumps({"n": n, "fwd_us": out}))
