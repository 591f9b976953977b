#!/usr/bin/env python
"""Would splitting the offsets of a SMALL, dense layer over several workgroups pay?  A strided 64 -> 128 layer on
the third level of the config-3 chain (26 k output rows, every tile walks all 27 offsets: 410 tiles x 27 steps of
~1.2 us on 256 CUs).  Times the forward with all offsets and with every second / third offset masked off -- what a
2- / 3-way split-K workgroup would run (the partial sums' reduction not included)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spconv_amd.pytorch import ops

dev = torch.device("cuda:0")
idx, shape = bench.make_scene("fixture", 0, 0)
ind = torch.from_numpy(idx).to(dev)
res = []
C = 16
for level, K in ((1, 32), (2, 64), (3, 128)):
    rb, out_shape = ops.build_rulebook(ind, 1, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, [0] * 3, False)
    f = torch.randn(rb.n_in, C, device=dev).half()
    w = (torch.randn(K, 3, 3, 3, C, device=dev) * 0.1).half()
    t = lambda fn: round(1e3 * bench.event_time_ms(fn, span=4), 1)
    row = dict(level=level, C=C, K=K, n_in=rb.n_in, n_out=rb.n_out)
    for name, keep in (("all", 0x7ffffff), ("half", 0x5555555), ("third", 0x1249249)):
        m = (rb.mask_fwd & keep).contiguous()
        row[name + "_us"] = t(lambda i: ops.igemm_fwd(f, w, rb.pair_fwd, m, None, rb.n_out, -1))
    res.append(row)
    ind, shape, C = rb.out_indices, out_shape, K
print(json.dumps(res))
