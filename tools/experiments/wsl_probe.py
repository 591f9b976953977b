#!/usr/bin/env python
"""Forward of the narrow layers of the config-4 backbone on its REAL levels (4 LiDAR-like scenes of 100 k voxels):
igemm_v4_kernel (SPX_WSL=0) against the weight-resident igemm_wsl_kernel (SPX_WSL=1, csrc/igemm_wsl.hip) --
SubM 16->16 on level 1, the strided 16->32 between levels 1 and 2, SubM 32->32 on level 2, the strided 32->64 behind
it; device time over hipGraph replays, bit-identity of the two results.
    python tools/wsl_probe.py        -> one JSON line per layer"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from spconv_amd import _lib
from spconv_amd.pytorch import ops
from spconv_amd.utils import nets

dev = torch.device("cuda:0")
L = _lib.load()
idx, shape = bench.make_scene("lidar", 100_000, 0, batch=4, shape=nets.SECOND_SHAPE)
ind = torch.from_numpy(idx).to(dev)
dtype = torch.bfloat16 if os.environ.get("WSL_BF16") else torch.float16


def run(name, f, w, pair, mask, n_out, ik, pairs):
    if os.environ.get("WSL_ONLY") and os.environ["WSL_ONLY"] not in name:
        return
    out = {"layer": name, "rows_out": n_out, "rows_in": f.shape[0], "C": f.shape[1], "K": w.shape[0],
           "pairs_per_row": round(pairs / n_out, 2)}
    res = {}
    for mode in (0, 1):
        L.spx_set_option(b"SPX_WSL", mode)
        fn = lambda i: ops.igemm_fwd(f, w, pair, mask, None, n_out, ik)
        res[mode] = fn(0).clone()
        if os.environ.get("WSL_EAGER"):          # (PMC passes: every launch a dispatch of its own)
            out["wsl_us" if mode else "v4_us"] = round(1e3 * bench.event_time_ms(fn, iters=10, warm=2, span=0), 1)
        else:
            out["wsl_us" if mode else "v4_us"] = round(1e3 * bench.event_time_ms(fn, span=4), 1)
    L.spx_set_option(b"SPX_WSL", -1)
    out["maxdiff"] = float((res[0].float() - res[1].float()).abs().max())
    out["identical"] = bool(torch.equal(res[0], res[1]))
    ab = n_out * w.shape[0] * 2 + f.shape[0] * f.shape[1] * 2 + 27 * n_out * 4 + n_out * 4
    out["alg_MB"] = round(ab / 1e6, 1)
    out["ideal_us_6TBps"] = round(ab / 6e6, 1)
    print(json.dumps(out), flush=True)


for level, C in ((1, 16), (2, 32)):
    n = ind.shape[0]
    rb = ops.build_rulebook(ind, 4, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)[0]
    num = rb.num_per_loc.cpu().numpy()
    f = torch.randn(n, C, device=dev).to(dtype)
    w = (torch.randn(C, 3, 3, 3, C, device=dev) * 0.1).to(dtype)
    run(f"subm{level} {C}->{C}", f, w, rb.pair_fwd, rb.mask_fwd, n, 13, int(n + 2 * num[:13].sum()))
    rb2, shape2 = ops.build_rulebook(ind, 4, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, [0] * 3, False)
    n2 = rb2.out_indices.shape[0]
    w2 = (torch.randn(2 * C, 3, 3, 3, C, device=dev) * 0.1).to(dtype)
    run(f"conv{level} {C}->{2 * C} s2", f, w2, rb2.pair_fwd, rb2.mask_fwd, n2, -1, int(rb2.num_per_loc.sum()))
    ind, shape = rb2.out_indices, shape2
