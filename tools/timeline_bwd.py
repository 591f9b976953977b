#!/usr/bin/env python
"""Per-workgroup timeline of the fused backward launch (igemm_bwd_kernel; debug build).

    SPX_LIB=spconv_amd/lib/libspconv_amd_dbg.so python tools/timeline_bwd.py [uniform|lidar]

wgrad workgroups come first in the grid (stamps: 0 entry, 1 first rows issued, 4 last chunk loop
done, 6 partial stores issued, 7 retired), dgrad tiles after them (stamps as tools/timeline.py).
s_memtime is per XCD, so every time is taken relative to the first entry on the same XCD
(workgroup b runs on XCD b % 8)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import SHAPE  # noqa: E402
from spconv_amd import _lib  # noqa: E402
from spconv_amd.pytorch import ops  # noqa: E402
from spconv_amd.utils import synthetic  # noqa: E402


def pct(a, qs=(0, 10, 50, 90, 100)):
    return [round(float(v), 2) for v in np.percentile(a, qs)]


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "uniform"
    dev = torch.device("cuda:0")
    n, C = 100000, 64
    gen = synthetic.uniform_scene if scene == "uniform" else synthetic.lidar_like_scene
    idx = torch.from_numpy(gen(SHAPE, n, 1, seed=0)).to(dev)
    n = idx.shape[0]
    f = (torch.rand(n, C, device=dev) * 2 - 1).half()
    dout = ((torch.rand(n, C, device=dev) * 2 - 1) * 0.2).half()
    w = (torch.rand(C, 3, 3, 3, C, device=dev) * 2 - 1).half()
    srt = len(sys.argv) > 2 and sys.argv[2] == "sort"
    rb, _ = ops.build_rulebook(idx, 1, SHAPE, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True, do_sort=srt)
    plan = ops._plan_of(rb)
    pair, mask, order_, to = ops.tables_of(rb, "fwd", C)
    L = _lib.load()
    L.spx_debug_timeline.restype = ctypes.c_int
    L.spx_debug_timeline.argtypes = [ctypes.c_void_p]

    def run():
        return ops.igemm_bwd(f, dout, w, pair, mask, order_, rb.pair_native,
                             rb.num_per_loc, True, plan, tile_order=to)
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    call_us = e0.elapsed_time(e1) / 20 * 1e3
    run()
    torch.cuda.synchronize()
    buf = np.zeros((8192, 8), dtype=np.uint64)
    _lib.check(L.spx_debug_timeline(buf.ctypes.data))
    ntiles = (n + 127) // 128
    # wgrad groups = blocks before the dgrad tiles (the library's sizing rule; TL_NW overrides)
    nw = int(os.environ.get("TL_NW", "0")) or (1024 - ntiles if 128 <= 1024 - ntiles < 384 else 384)
    tick = 1.0 / float(os.environ.get("SPX_TICK_MHZ", "2200"))   # s_memtime: shader clock under load (~2.2 GHz)
    t = buf[:nw + ntiles].astype(np.int64)
    # s_memtime is per XCD and the counters are not aligned: group the workgroups by clock domain
    # (entries of one launch lie within a few thousand ticks, the domains are millions apart) and
    # measure every stamp from the first entry of its domain
    order = np.argsort(t[:, 0])
    gaps = np.diff(t[order, 0]) > 200000
    domain = np.zeros(t.shape[0], dtype=np.int64)
    domain[order] = np.concatenate([[0], np.cumsum(gaps)])
    rel = np.zeros(t.shape, dtype=np.float64)
    for d in range(int(domain.max()) + 1):
        sel = domain == d
        rel[sel] = (t[sel] - t[sel, 0].min()) * tick
    xcd_of = domain
    wg, dg = rel[:nw], rel[nw:]
    out = {"scene": scene, "n": int(n), "call_us_events": round(call_us, 2), "wgrad_groups": nw,
           "dgrad_tiles": int(ntiles),
           "wgrad": {"entry": pct(wg[:, 0]), "rows_issued": pct(wg[:, 1]), "loop_done": pct(wg[:, 4]),
                     "stores_issued": pct(wg[:, 6]), "retired": pct(wg[:, 7]),
                     "lifetime": pct(wg[:, 7] - wg[:, 0])},
           "dgrad": {"entry": pct(dg[:, 0]), "mask_known": pct(dg[:, 2]), "prologue_done": pct(dg[:, 3]),
                     "loop_done": pct(dg[:, 4]), "stores_issued": pct(dg[:, 6]), "retired": pct(dg[:, 7]),
                     "lifetime": pct(dg[:, 7] - dg[:, 0])},
           "clock_domains": int(domain.max()) + 1,
           "kernel_span_per_domain": [round(float(rel[domain == d][:, 7].max()), 2) for d in range(int(domain.max()) + 1)]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
