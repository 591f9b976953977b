#!/usr/bin/env python
"""Where does the dense-scene forward spend its time?  Times the gather-GEMM on the reference fixture scene
(C = K = 64, fp16) -- run once per ablation build (csrc/build_ablate.sh):
   for v in 0 1 2 3 4 5; do SPX_LIB=spconv_amd/lib/libspconv_amd_abl$v.so python tools/dense_probe.py; done
(The halo-kernel half of this probe went with the kernel in round 4: profiles/r02_dense_regime_experiments.md.)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from spconv_amd.pytorch import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    C = int(os.environ.get("PROBE_C", "64"))
    kind = os.environ.get("PROBE_SCENE", "fixture")
    idx, shape = bench.make_scene(kind, 100_000, 0)
    ind = torch.from_numpy(idx).to(dev)
    rb = ops.build_rulebook(ind, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)[0]
    n = idx.shape[0]
    f = torch.randn(n, C, device=dev).half()
    w = (torch.randn(C, 3, 3, 3, C, device=dev) * 0.1).half()
    res = {"lib": os.environ.get("SPX_LIB", "default"), "C": C}
    span = 0 if os.environ.get("PROBE_EAGER") == "1" else 8
    res["v4_fwd_us"] = round(1e3 * bench.event_time_ms(
        lambda i: ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, n, 13), span=span), 2)
    res["v4_dgrad_us"] = round(1e3 * bench.event_time_ms(
        lambda i: ops.igemm_dgrad(f, w, rb.pair_fwd, rb.mask_fwd, None, n, True), span=span), 2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
