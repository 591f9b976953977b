#!/usr/bin/env python
"""Per-workgroup timeline of igemm_v4_kernel (debug build, spconv_amd/csrc/build_debug.sh).

    SPX_LIB=spconv_amd/lib/libspconv_amd_dbg.so python tools/timeline.py [uniform|lidar] [centre|sort|i8|i8sort|bwd]

(sort = the modules' default rows layout; i8 = the int8 layer of config 5; bwd = the fused backward launch, wgrad ranges |
appendix workgroups | dgrad tiles, under the rows layout; TL_RAW=<file.npy> keeps the raw stamp table.)

Every workgroup stamps s_memtime (100 MHz constant clock on gfx950: 10 ns ticks) at
0 entry, 1 identity loads issued, 2 tile mask known, 3 prologue done, 4 main loop done,
5 accumulators staged, 6 stores issued, 7 stores retired.  Prints percentiles of each stamp
relative to the first workgroup's entry, and of the phase durations."""
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


def bwd_timeline(scene, idx, f, w, n, C):
    """The fused backward launch (igemm_bwd_kernel): [wgrad ranges | appendix workgroups | main dgrad tiles] under the
    modules' default rows layout.  wgrad workgroups stamp 0 entry, 1 first rows issued, 4 last chunk loop done,
    6 partials issued, 7 retired; dgrad tiles as in the forward."""
    rb, _ = ops.build_rulebook(idx, 1, SHAPE, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True, do_sort="layout")
    dout = (torch.rand(n, C, device=f.device) * 2 - 1).half()
    pair, mask, order, to = ops.tables_of(rb, "fwd", C)
    plan = ops._plan_of(rb)
    L = _lib.load()
    getter = L.spx_debug_timeline
    getter.restype = ctypes.c_int
    getter.argtypes = [ctypes.c_void_p]
    for _ in range(20):
        ops.igemm_bwd(f, dout, w, pair, mask, order, rb.pair_native, rb.num_per_loc, True, plan, tile_order=to)
    torch.cuda.synchronize()
    buf = np.zeros((8192, 8), dtype=np.uint64)
    _lib.check(getter(buf.ctypes.data))
    tick = 1.0 / float(os.environ.get("SPX_TICK_MHZ", "2200"))
    if os.environ.get("TL_RAW"):
        np.save(os.environ["TL_RAW"], buf)
    a = buf.astype(np.int64)
    live = a[:, 0] > 0
    last = int(np.nonzero(live)[0].max()) + 1
    a = a[:last]
    ntiles = (n + 127) // 128
    napp = last - ntiles
    # wgrad workgroups never stamp 2 (tile mask known); the appendix workgroups that left at once stamp 0 only
    nw = 0
    while nw < last and a[nw, 2] == 0 and a[nw, 7] > 0:
        nw += 1
    napp -= nw
    wg, ap, mt = a[:nw], a[nw:nw + napp], a[nw + napp:]
    ap_live = ap[ap[:, 7] > 0]
    out = {"scene": scene, "mode": "fused backward", "workgroups": last, "wgrad": nw, "appendix": int(napp),
           "appendix_with_rows": int(ap_live.shape[0]), "main_tiles": int(mt.shape[0])}
    # s_memtime is per XCD: every figure is a difference taken inside one XCD (workgroup b runs on XCD b % 8)
    t0x = np.zeros(8)
    for x in range(8):
        g = a[x::8]
        g = g[g[:, 7] > 0]
        t0x[x] = g[:, 0].min()
    def rel(block_rows, first_index):
        idxs = first_index + np.arange(block_rows.shape[0])
        return (block_rows - t0x[idxs % 8][:, None]) * tick
    for nm, rows, first in (("wgrad", wg, 0), ("appendix", ap, nw), ("main", mt, nw + napp)):
        keep = rows[:, 7] > 0
        if not keep.any():
            continue
        r = rel(rows, first)[keep]
        out[nm] = {"entry_us": [round(float(v), 2) for v in np.percentile(r[:, 0], [0, 50, 100])],
                   "retire_us": [round(float(v), 2) for v in np.percentile(r[:, 7], [0, 50, 90, 100])],
                   "lifetime_us": [round(float(v), 2) for v in np.percentile(r[:, 7] - r[:, 0], [10, 50, 90, 100])]}
    spans = []
    for x in range(8):
        g = a[x::8]
        g = g[g[:, 7] > 0]
        spans.append((g[:, 7].max() - g[:, 0].min()) * tick)
    out["xcd_span_us"] = [round(float(v), 2) for v in spans]
    print(json.dumps(out))


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "uniform"
    centre = len(sys.argv) > 2 and sys.argv[2] == "centre"
    srt = len(sys.argv) > 2 and sys.argv[2] in ("sort", "i8sort") and "layout"   # the modules' default rows layout
    i8 = len(sys.argv) > 2 and sys.argv[2] in ("i8", "i8sort")      # BASELINE config 5: int8, C = K = 128, 200 k voxels
    dev = torch.device("cuda:0")
    n, C = (200000, 128) if (len(sys.argv) > 2 and sys.argv[2] in ("i8", "i8sort")) else (100000, 64)
    n = int(os.environ.get("TL_N", n))                     # voxels (the stamp table holds 8192 workgroups)
    C = int(os.environ.get("TL_C", C))
    gen = synthetic.uniform_scene if scene == "uniform" else synthetic.lidar_like_scene
    idx = torch.from_numpy(gen(SHAPE, n, 1, seed=0)).to(dev)
    f = (torch.rand(n, C, device=dev) * 2 - 1).half()
    w = (torch.rand(C, 3, 3, 3, C, device=dev) * 2 - 1).half()
    rb, _ = ops.build_rulebook(idx, 1, SHAPE, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True, do_sort=srt)
    mask = torch.full_like(rb.mask_fwd, 1 << 13) if centre else rb.mask_fwd
    pair, order, to = rb.pair_fwd, None, 0
    if srt:
        pair, mask, order, to = ops.tables_of(rb, "fwd", C)
    if i8:
        f = torch.randint(-127, 128, (n, C), dtype=torch.int8, device=dev)
        w = torch.randint(-127, 128, (C, 3, 3, 3, C), dtype=torch.int8, device=dev)
        sc = torch.rand(C, device=dev) * 1e-2
        bi = torch.rand(C, device=dev)
    if len(sys.argv) > 2 and sys.argv[2] == "bwd":
        return bwd_timeline(scene, idx, f, w, n, C)
    L = _lib.load()
    getter = L.spx_debug_timeline
    getter.restype = ctypes.c_int
    getter.argtypes = [ctypes.c_void_p]
    for _ in range(20):
        (ops.igemm_fwd_int8(f, w, pair, mask, order, n, 13, sc, bi, None, 0.0, torch.int8, ops.Activation.ReLU, 0.0,
                            tile_order=to) if i8 else ops.igemm_fwd(f, w, pair, mask, order, n, 13, tile_order=to))
    torch.cuda.synchronize()
    (ops.igemm_fwd_int8(f, w, pair, mask, order, n, 13, sc, bi, None, 0.0, torch.int8, ops.Activation.ReLU, 0.0,
                            tile_order=to) if i8 else ops.igemm_fwd(f, w, pair, mask, order, n, 13, tile_order=to))
    buf = np.zeros((8192, 8), dtype=np.uint64)
    _lib.check(getter(buf.ctypes.data))
    mb = 2 if n > 32 * 1024 else 1                 # the library's tile-height rule (csrc/igemm.hip dispatch_gather_gemm)
    ntiles = (n + 64 * mb - 1) // (64 * mb)
    if i8:
        ntiles = (n + 127) // 128
    napp = 0
    if srt:                       # rows layout: the appendix workgroups lead the grid (csrc/igemm.hip)
        napp = (n // 4 + 64 * mb - 1) // (64 * mb)
    allt = buf[:napp + ntiles].astype(np.int64)
    app = allt[:napp]
    real_app = app[app[:, 7] > 0]
    t = allt[napp:]
    t0 = allt[:, 0].min()
    rel = (t - t0) / float(os.environ.get("SPX_TICK_MHZ", "2200"))       # s_memtime: shader clock under load
    names = ["entry", "ident_issued", "mask_known", "prologue_done", "loop_done", "staged",
             "stores_issued", "stores_retired"]
    out = {"scene": scene, "centre_only": centre, "tiles": int(ntiles), "appendix_workgroups": int(napp),
           "appendix_with_rows": int(real_app.shape[0]), "stamps_us": {}, "phases_us": {}}
    if real_app.shape[0]:
        ra = (real_app - t0) / float(os.environ.get("SPX_TICK_MHZ", "2200"))
        out["appendix_stamps_us"] = {nm: [round(float(v), 2) for v in np.percentile(ra[:, i], [0, 50, 100])]
                                     for i, nm in enumerate(names) if i != 5}
        out["appendix_lifetime_us"] = [round(float(v), 2) for v in np.percentile(ra[:, 7] - ra[:, 0], [10, 50, 90, 100])]
        out["launch_entry_to_last_retire_us"] = round(float((allt[:, 7].max() - t0) / float(os.environ.get("SPX_TICK_MHZ", "2200"))), 2)
    used = [0, 1, 2, 3, 4, 6, 7]                    # (stamp 5 is not taken by the current kernel)
    for i in used:
        q = np.percentile(rel[:, i], [0, 10, 50, 90, 100])
        out["stamps_us"][names[i]] = [round(float(v), 2) for v in q]
    for a, b in zip(used[:-1], used[1:]):
        d = rel[:, b] - rel[:, a]
        q = np.percentile(d, [10, 50, 90, 100])
        out["phases_us"][f"{names[a]}->{names[b]}"] = [round(float(v), 2) for v in q]
    # s_memtime is per XCD (not synchronised across dies): spans are taken inside each XCD
    # (workgroup b runs on XCD b % 8) and the stamps above are only meaningful as differences
    tick = 1.0 / float(os.environ.get("SPX_TICK_MHZ", "2200"))
    spans, ramps = [], []
    for x in range(8):
        g = allt[x::8]
        g = g[g[:, 7] > 0]
        spans.append((g[:, 7].max() - g[:, 0].min()) * tick)
        ramps.append((g[:, 0].max() - g[:, 0].min()) * tick)
    out["xcd_span_us"] = [round(float(v), 2) for v in spans]
    out["xcd_entry_ramp_us"] = [round(float(v), 2) for v in ramps]
    out["wg_lifetime_us"] = [round(float(v), 2) for v in np.percentile((t[:, 7] - t[:, 0]) * tick, [10, 50, 90, 100])]
    # how well the launch packs: busy time of all workgroups / (slots x the longest per-XCD span); the rest is
    # ramp, dispatch rounds that end with a heavy tile, and the tail
    slots = int(os.environ.get("TL_SLOTS", "1024"))
    busy = float(((t[:, 7] - t[:, 0]) * tick).sum())
    out["packing"] = {"slots": slots, "busy_us_total": round(busy, 1),
                      "span_us": round(float(np.median(spans)), 2),
                      "efficiency": round(busy / (min(slots, ntiles) * float(np.median(spans))), 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
