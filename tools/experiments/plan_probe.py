#!/usr/bin/env python
"""Device time of the wgrad plan launch (one block) at the backbone's layer sizes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spconv_amd.pytorch import ops
dev = torch.device("cuda:0")
for n, kv, subm in ((125_000, 27, 1), (400_000, 27, 1), (125_000, 27, 0), (30_000, 27, 0)):
    num = torch.randint(n // 8, n // 2, (kv,), dtype=torch.int32, device=dev)
    for _ in range(5):
        ops.wgrad_plan(num, n, kv, bool(subm))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.wgrad_plan(num, n, kv, bool(subm))
    e1.record(); torch.cuda.synchronize()
    print(f"n {n} kv {kv} subm {subm}: {e0.elapsed_time(e1) / 200 * 1e3:.2f} us per plan (incl. launch gaps)")
