#!/usr/bin/env python
"""Per-kernel device times (HIP events) for the hot path over a small scene/sort matrix.
Used for A/B tuning runs on the GPU box: `SPX_GEMM_DEPTH=2 python tools/kbench.py`."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import SHAPE, event_time_ms  # noqa: E402
from spconv_amd.pytorch import ops  # noqa: E402
from spconv_amd.utils import synthetic  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    C = K = int(os.environ.get("KB_CHANNELS", "64"))
    n = int(os.environ.get("KB_VOXELS", "100000"))
    tag = {k: v for k, v in os.environ.items() if k.startswith("SPX_")}
    rows = []
    # reference floors on this box: a plain 12.8 MB -> 12.8 MB copy and the dense centre GEMM
    ff = (torch.rand(n, C, device=dev) * 2 - 1).half()
    oo = torch.empty_like(ff)
    ww = (torch.rand(C, K, device=dev) * 2 - 1).half()
    floors = dict(copy_us=round(event_time_ms(lambda: oo.copy_(ff)) * 1e3, 2),
                  mm_us=round(event_time_ms(lambda: torch.mm(ff, ww, out=oo)) * 1e3, 2),
                  empty_launch_us=round(event_time_ms(lambda: oo[:64].zero_()) * 1e3, 2))
    for scene in os.environ.get("KB_SCENES", "uniform,lidar").split(","):
        gen = synthetic.uniform_scene if scene == "uniform" else synthetic.lidar_like_scene
        idx = torch.from_numpy(gen(SHAPE, n, 1, seed=0)).to(dev)
        nn = idx.shape[0]
        f = (torch.rand(nn, C, device=dev) * 2 - 1).half()
        d = ((torch.rand(nn, K, device=dev) * 2 - 1) * 0.2).half()
        w = (torch.rand(K, 3, 3, 3, C, device=dev) * 2 - 1).half()
        for sort in ((False, True) if os.environ.get("KB_SORT", "1") == "1" else (False,)):
            rb, _ = ops.build_rulebook(idx, 1, SHAPE, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True,
                                       do_sort=sort)
            plan = ops._plan_of(rb)
            t_f = event_time_ms(lambda: ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, rb.argsort_fwd, nn, 13))
            t_d = event_time_ms(lambda: ops.igemm_dgrad(d, w, rb.pair_fwd, rb.mask_fwd, rb.argsort_fwd, nn, True))
            t_w = event_time_ms(lambda: ops.igemm_wgrad(f, d, w.shape, rb.pair_native, rb.num_per_loc, True, plan))
            if not sort:
                # centre-only mask: cost of the kernels without any non-identity step
                m1 = torch.full_like(rb.mask_fwd, 1 << 13)
                rowx = dict(fwd_centre_only_us=round(event_time_ms(
                    lambda: ops.igemm_fwd(f, w, rb.pair_fwd, m1, None, nn, 13)) * 1e3, 2))
            else:
                rowx = {}
            t_r = event_time_ms(lambda: ops.build_rulebook(idx, 1, SHAPE, [3] * 3, [1] * 3, [1] * 3, [1] * 3,
                                                           [0] * 3, True, do_sort=sort), iters=5, warm=2)
            rows.append(dict(scene=scene, sort=sort, fwd_us=round(t_f * 1e3, 2), dgrad_us=round(t_d * 1e3, 2),
                             wgrad_us=round(t_w * 1e3, 2), rulebook_us=round(t_r * 1e3, 1), **rowx))
    # BASELINE config 5: int8 SubMConv3d C = K = 128, 200k voxels, per-channel scale + ReLU, int8 out
    cfg5 = {}
    if os.environ.get("KB_INT8", "1") == "1":
        n5, C5 = 200_000, 128
        idx = torch.from_numpy(synthetic.uniform_scene(SHAPE, n5, 1, seed=0)).to(dev)
        rb, _ = ops.build_rulebook(idx, 1, SHAPE, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)
        f8 = torch.randint(-127, 128, (n5, C5), dtype=torch.int8, device=dev)
        w8 = torch.randint(-127, 128, (C5, 3, 3, 3, C5), dtype=torch.int8, device=dev)
        sc = torch.rand(C5, device=dev) * 1e-3 + 5e-4
        bs = torch.rand(C5, device=dev) * 10 - 5
        t_i8 = event_time_ms(lambda: ops.igemm_fwd_int8(f8, w8, rb.pair_fwd, rb.mask_fwd, None, n5, 13, sc, bs,
                                                        None, 0.0, torch.int8, ops.Activation.ReLU))
        fh = (torch.rand(n5, C5, device=dev) * 2 - 1).half()
        wh = (torch.rand(C5, 3, 3, 3, C5, device=dev) * 2 - 1).half()
        t_h = event_time_ms(lambda: ops.igemm_fwd(fh, wh, rb.pair_fwd, rb.mask_fwd, None, n5, 13))
        alg = n5 * (C5 + C5) + 4 * 27 * n5 + 27 * C5 * C5       # algorithmic bytes
        cfg5 = dict(n=n5, C=C5, int8_fwd_us=round(t_i8 * 1e3, 2), f16_fwd_us=round(t_h * 1e3, 2),
                    int8_Gvox_s=round(n5 / t_i8 / 1e6, 3), int8_alg_GBps=round(alg / t_i8 / 1e6, 1))
    print(json.dumps({"env": tag, "C": C, "n": n, "floors": floors, "rows": rows, "cfg5_int8": cfg5}))


if __name__ == "__main__":
    main()
