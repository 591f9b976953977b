#!/usr/bin/env python
"""A/B of the dense-neighbourhood gather-GEMM (csrc/igemm_ns.hip, SPX_GEMM_NS=1) against igemm_v4_kernel: results and
device time (hipGraph replays between HIP events) of the forward on the benchmark scenes.   python tools/ns_probe.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from spconv_amd import _lib  # noqa: E402
from spconv_amd.pytorch import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    L = _lib.load()
    rows = []
    for kind, n, C in (("fixture", 0, 64), ("lidar", 100_000, 64), ("uniform", 100_000, 64), ("fixture", 0, 32), ("lidar", 300_000, 64)):
        idx, shape = bench.make_scene(kind, n, 0)
        ind = torch.from_numpy(idx).to(dev)
        rb = ops.build_rulebook(ind, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)[0]
        N = ind.shape[0]
        g = torch.Generator(device="cpu").manual_seed(1)
        f = (torch.rand((N, C), generator=g) * 2 - 1).to(dev).half()
        w = (torch.rand((64, 3, 3, 3, C), generator=g) * 2 - 1).to(dev).half()
        fwd = lambda i: ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, N, 13)
        row = dict(scene=kind, voxels=N, C=C, pairs=int(N + 2 * rb.num_per_loc[:13].sum().item()))
        _lib.check(L.spx_set_option(b"SPX_GEMM_NS", 0))
        ref = fwd(0)
        row["v4_us"] = round(bench.event_time_ms(fwd, span=8) * 1e3, 2)
        _lib.check(L.spx_set_option(b"SPX_GEMM_NS", 1))
        got = fwd(0)
        torch.cuda.synchronize()
        row["ns_us"] = round(bench.event_time_ms(fwd, span=8) * 1e3, 2)
        row["bit_identical"] = bool(torch.equal(ref, got))
        row["max_abs_diff"] = float((ref.float() - got.float()).abs().max())
        row["max_abs_ref"] = float(ref.float().abs().max())
        got2 = fwd(0)
        row["reproducible"] = bool(torch.equal(got, got2))
        _lib.check(L.spx_set_option(b"SPX_GEMM_NS", 0))
        rows.append(row)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
