#!/usr/bin/env python
"""Where does the dense-scene forward spend its time?  Times the plain and the halo gather-GEMM on the
reference fixture scene (C = K = 64, fp16) -- run once per SPX_HALO_DBG value (the env is read once
per process):   for d in 0 1 2 3 4; do SPX_HALO_DBG=$d python tools/dense_probe.py; done"""
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
    ops._TILE_MODE = "1"
    tp = ops.tile_plan(rb, "fwd")
    n = idx.shape[0]
    f = torch.randn(n, C, device=dev).half()
    w = (torch.randn(C, 3, 3, 3, C, device=dev) * 0.1).half()
    res = {"dbg": os.environ.get("SPX_HALO_DBG", "0"), "v4dbg": os.environ.get("SPX_V4_DBG", "0"), "C": C}
    try:
        import ctypes
        from spconv_amd import _lib
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        _lib.load().spx_debug_occupancy(ctypes.byref(a), ctypes.byref(b))
        res["occupancy_wgs_per_cu"] = {"halo": a.value, "v4": b.value}
    except AttributeError:
        pass
    span = 0 if os.environ.get("PROBE_EAGER") == "1" else 8
    res["v4_fwd_us"] = round(1e3 * bench.event_time_ms(
        lambda i: ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, n, 13), span=span), 2)
    res["v4_dgrad_us"] = round(1e3 * bench.event_time_ms(
        lambda i: ops.igemm_dgrad(f, w, rb.pair_fwd, rb.mask_fwd, None, n, True), span=span), 2)
    if os.environ.get("PROBE_HALO", "0") != "1":
        print(json.dumps(res))
        return
    res["halo_fwd_us"] = round(1e3 * bench.event_time_ms(
        lambda i: ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, n, 13, plan=tp), span=span), 2)
    res["halo_dgrad_us"] = round(1e3 * bench.event_time_ms(
        lambda i: ops.igemm_dgrad(f, w, rb.pair_fwd, rb.mask_fwd, None, n, True, plan=tp), span=span), 2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
