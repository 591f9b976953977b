"""A/B timing of the gather-GEMM generations (SPX_GEMM_V = 4 | 5) on the benchmark scenes:
forward and dgrad device time per launch (hipGraph replay over rotating scenes, HIP events), plus a
bit-identity check.  Kernel-variant switches that the library caches per process (SPX_V5_VARIANT,
SPX_V5_WGS, SPX_V5_PAIR16) come from the environment of the call.

    python tools/v5_bench.py [--scene uniform|fixture] [--voxels N] [--channels C] [--scenes S]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from spconv_amd import _lib  # noqa: E402
from spconv_amd.pytorch import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="uniform")
    ap.add_argument("--voxels", type=int, default=100_000)
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--scenes", type=int, default=8)
    ap.add_argument("--dtype", default="f16")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dtype = {"f16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
    C = K = a.channels
    L = _lib.load()
    scenes = []
    for si in range(a.scenes):
        idx, shape = bench.make_scene(a.scene, a.voxels, seed=si)
        ind = torch.from_numpy(idx).to(dev)
        rb = ops.build_rulebook(ind, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)[0]
        g = torch.Generator(device="cpu").manual_seed(si)
        f = (torch.rand((idx.shape[0], C), generator=g) * 2 - 1).to(dev, dtype)
        d = ((torch.rand((idx.shape[0], K), generator=g) * 2 - 1) * 0.2).to(dev, dtype)
        scenes.append((rb, f, d))
    w = (torch.rand((K, 3, 3, 3, C), generator=torch.Generator().manual_seed(99)) * 2 - 1).to(dev, dtype)
    n = scenes[0][0].n_in
    S = len(scenes)

    def fwd(i):
        rb, f, _ = scenes[i % S]
        return ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, 13)

    def dgrad(i):
        rb, _, d = scenes[i % S]
        return ops.igemm_dgrad(d, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_in, True)

    res = {"scene": a.scene, "voxels": n, "C": C, "dtype": a.dtype,
           "env": {k: v for k, v in os.environ.items() if k.startswith("SPX_")}}
    outs = {}
    for v in (4, 6):
        _lib.check(L.spx_set_option(b"SPX_GEMM_V", v))
        outs[v] = (fwd(0), dgrad(0))
        torch.cuda.synchronize()
        res[f"v{v}"] = {"fwd_us": round(bench.event_time_ms(fwd, span=max(S, 8)) * 1e3, 2),
                        "dgrad_us": round(bench.event_time_ms(dgrad, span=max(S, 8)) * 1e3, 2)}
    res["fwd_bit_identical"] = bool(torch.equal(outs[4][0], outs[6][0]))
    res["dgrad_bit_identical"] = bool(torch.equal(outs[4][1], outs[6][1]))
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
