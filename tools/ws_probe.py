#!/usr/bin/env python
"""Weight-stationary gather-GEMM (csrc/igemm_ws.hip) against igemm_v4 on dense scenes: correctness (against v4 and, on a
row sample, against an fp32 torch reference) and device time per launch shape, rows in rulebook order and mask-sorted.
    python tools/ws_probe.py [fixture|lidar]        (WS_TL=1: per-wave timeline, WS_PROBE_SORTED=1: sorted orders too)
Prints one JSON line per measurement."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from spconv_amd import _lib  # noqa: E402
from spconv_amd.pytorch import ops  # noqa: E402


def set_ws(v):
    _lib.check(_lib.load().spx_set_option(b"SPX_WS", int(v)))


def ref_rows(f, w, pair, rows, dgrad=False):
    """fp32 reference of the output rows `rows` (forward: out[o] = sum_k f[pair[k][o]] W[:, k, :]^T)."""
    K, C = w.shape[0], w.shape[-1]
    kv = pair.shape[0]
    w32 = w.float().reshape(K, kv, C)
    out = torch.zeros(len(rows), C if dgrad else K, device=f.device)
    for k in range(kv):
        idx = pair[k][rows].long()
        ok = idx >= 0
        g = f[idx.clamp(min=0)].float() * ok[:, None]
        kb = kv - 1 - k if dgrad else k
        out += g @ (w32[:, kb, :] if dgrad else w32[:, kb, :].t())
    return out


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "fixture"
    variants = [1]              # SPX_WS = 1: the weight-stationary kernel wherever its shape limits allow
    dev = torch.device("cuda:0")
    idx, shape = bench.make_scene("fixture" if kind == "fixture" else "lidar", 110_000, 0)
    ind = torch.from_numpy(idx).to(dev)
    rb = ops.build_rulebook(ind, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)[0]
    n = idx.shape[0]
    torch.manual_seed(0)
    f = torch.randn(n, 64, device=dev).half()
    w = (torch.randn(64, 3, 3, 3, 64, device=dev) * 0.1).half()
    print(json.dumps({"scene": kind, "n": n, "pairs_per_voxel": round(float((rb.pair_fwd >= 0).sum().item()) / n, 3)}))
    sample = torch.randperm(n, device=dev)[:4096]
    pair0, mask0 = rb.pair_fwd, rb.mask_fwd
    ops.sort_rulebook(rb)
    pair1, mask1 = rb.sorted_tables["fwd"]
    order = rb.argsort_fwd
    orders = {"rows": (pair0, mask0, None, 0), "sorted": (pair1, mask1, order, 1)}
    if os.environ.get("WS_PROBE_SORTED", "0") != "1":
        del orders["sorted"]
    # mask-sorted 32-row groups dealt round-robin to the workgroups (group g of the sorted order -> workgroup g % ntiles,
    # wave g // ntiles): every tile gets light and heavy groups, so the launch does not end with the heaviest tile
    L = _lib.load()
    for tm in (256, 512):
        gpt = tm // 32
        ngroups = (n + 31) // 32
        ntiles = (ngroups + gpt - 1) // gpt
        g = torch.arange(ntiles * gpt, device=dev)
        src_group = (g % gpt) * ntiles + (g // gpt)                       # position group g reads sorted group src_group
        rows = (src_group[:, None] * 32 + torch.arange(32, device=dev)[None, :]).reshape(-1)
        rows = rows[rows < n]
        assert rows.numel() == n and rows.unique().numel() == n
        # positions past the real rows must stay at the end: only valid when every group but the last is full --
        # take the simple route: order over the first n positions
        bal = order[rows.long()].contiguous()
        pair_b, mask_b = torch.empty_like(pair0), torch.empty_like(mask0)
        _lib.check(L.spx_permute_tables(pair0.data_ptr(), mask0.data_ptr(), bal.data_ptr(), pair0.shape[0], pair0.shape[1],
                                        mask0.shape[1], pair_b.data_ptr(), mask_b.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream))
        orders[f"balanced{tm}"] = (pair_b, mask_b, bal, 1)
    # every pair redirected to the row itself: the same instruction stream with perfect locality (the non-memory floor)
    rown = torch.arange(n, device=dev, dtype=torch.int32)[None, :].expand_as(pair0)
    pair_local = torch.where(pair0 >= 0, rown, pair0).contiguous()
    local = (pair_local, mask0, None, 0)
    refs = {"fwd": ref_rows(f, w, pair0, sample), "dgrad": ref_rows(f, w, pair0, sample, dgrad=True)}

    def run(which, tbl):
        pair, mask, arg, to = tbl
        if which == "fwd":
            return ops.igemm_fwd(f, w, pair, mask, arg, n, 13, tile_order=to)
        return ops.igemm_dgrad(f, w, pair, mask, arg, n, True, tile_order=to)

    for which in ("fwd", "dgrad"):
        base = {}
        for oname, tbl in orders.items():
            set_ws(-1)
            o4 = run(which, tbl)
            torch.cuda.synchronize()
            base[oname] = o4
            t = bench.event_time_ms(lambda i: run(which, tbl), span=8)
            err = float((o4[sample].float() - refs[which]).abs().max() / refs[which].abs().max())
            print(json.dumps({"kernel": "v4", "op": which, "order": oname, "us": round(1e3 * t, 2), "err_vs_fp32": err}))
        t = bench.event_time_ms(lambda i: run(which, local), span=8)
        print(json.dumps({"kernel": "v4", "op": which, "order": "local-pairs", "us": round(1e3 * t, 2)}))
        for v in variants:
            for oname, tbl in orders.items():
                set_ws(v)
                try:
                    o = run(which, tbl)
                    torch.cuda.synchronize()
                    d = (o.float() - base["rows"].float()).abs()
                    err = float((o[sample].float() - refs[which]).abs().max() / refs[which].abs().max())
                    bad = int((d > 2e-2 * base["rows"].float().abs().max()).sum().item())
                    t = bench.event_time_ms(lambda i: run(which, tbl), span=8)
                    print(json.dumps({"kernel": f"ws{v}", "op": which, "order": oname, "us": round(1e3 * t, 2),
                                      "err_vs_fp32": err, "maxdiff_vs_v4": float(d.max()), "bad": bad,
                                      "nan": int(torch.isnan(o.float()).sum().item())}))
                except Exception as e:                                   # a failing variant must not cost the others
                    print(json.dumps({"kernel": f"ws{v}", "op": which, "order": oname, "error": str(e)[:300]}))
                finally:
                    set_ws(-1)
            if os.environ.get("WS_TL") == "1":
                for oname, tbl in orders.items():
                    set_ws(v)
                    print(json.dumps({"kernel": f"ws{v}", "op": which, "order": oname, "timeline_ticks": timeline(lambda: run(which, tbl))}))
                    set_ws(-1)
            set_ws(v)
            t = bench.event_time_ms(lambda i: run(which, local), span=8)
            set_ws(-1)
            print(json.dumps({"kernel": f"ws{v}", "op": which, "order": "local-pairs", "us": round(1e3 * t, 2)}))


def timeline(run, nwaves_cap=1 << 16):
    """Per-wave stamps of one launch: percentiles of the intervals, in microseconds at 100 MHz-agnostic ticks."""
    import ctypes
    import numpy as np
    L = _lib.load()
    L.spx_debug_ws_timeline.argtypes = [ctypes.c_void_p]
    buf = torch.zeros(nwaves_cap * 8, dtype=torch.int64, device="cuda:0")
    run()
    torch.cuda.synchronize()
    L.spx_debug_ws_timeline(buf.data_ptr())
    run()
    torch.cuda.synchronize()
    L.spx_debug_ws_timeline(None)
    t = buf.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 0] > 0]
    names = ["entry->masks", "->phase0 staged", "->own phase0 steps done", "->all waves done", "->phase1 staged",
             "->walk done", "->stores issued"]
    seq = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7)]
    out = {"waves": int(t.shape[0])}
    for nm, (a, b) in zip(names, seq):
        ok = (t[:, a] > 0) & (t[:, b] > 0)
        d = (t[ok, b] - t[ok, a]).astype(np.float64)
        if d.size:
            out[nm] = [round(float(np.percentile(d, q)), 0) for q in (10, 50, 90, 100)]
    ok = t[:, 7] > 0
    out["life"] = [round(float(np.percentile((t[ok, 7] - t[ok, 0]).astype(np.float64), q)), 0) for q in (10, 50, 90, 100)]
    out["launch_span"] = float(t[ok, 7].max() - t[ok, 0].min())
    return out


if __name__ == "__main__":
    main()
