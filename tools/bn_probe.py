#!/usr/bin/env python
"""Device time of one fused BatchNorm1d + ReLU forward and backward (csrc/norm.hip, through the C ABI on preallocated
buffers, hipGraph replays of 20 calls) at the level shapes of the config-4 backbone, next to the time the
passes' bytes would take at 6 TB/s (tools/experiments/bn_two_launch.patch: the two-launch forms this tool compared).
    python tools/bn_probe.py            -> one JSON line per shape
Each launch alone: csrc/build_bn_probe.sh, then SPX_LIB=.../libspconv_amd_bnprobe.so SPX_BN_PHASES=1|2|4 (statistics pass |
merge | apply).  fwd_from_conv_records_us: merge + apply over the records a convolution's epilogue leaves (one per 128 rows)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spconv_amd import _lib

dev = torch.device("cuda:0")
L = _lib.load()
SHAPES = [(400_000, 16), (313_000, 32), (140_000, 64), (50_000, 64), (20_000, 128)]
F16, F32 = _lib.DTYPE_F16, _lib.DTYPE_F32


def timed(fn, s, reps=20, rounds=5):
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(reps):
            fn()
    best = 1e9
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            a.record(s); g.replay(); b.record(s)
        s.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / reps)
    return best


s = torch.cuda.Stream()
for n, C in SHAPES:
    x = torch.randn(n, C, device=dev).half()
    dy = torch.randn(n, C, device=dev).half()
    y, dx = torch.empty_like(x), torch.empty_like(x)
    w, b = torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) - 0.5
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    stats = torch.empty(2, C, device=dev)
    dw, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    ws = torch.empty(L.spx_batchnorm_ws_bytes(n, C), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    raw = s.cuda_stream

    def fwd():
        _lib.check(L.spx_batchnorm_fwd(x.data_ptr(), y.data_ptr(), n, C, F16, w.data_ptr(), b.data_ptr(), rm.data_ptr(),
                                       rv.data_ptr(), None, F32, 1, 0.01, 1e-3, 1, stats[0].data_ptr(), stats[1].data_ptr(),
                                       ws.data_ptr(), ws.numel(), None, raw))

    def bwd():
        _lib.check(L.spx_batchnorm_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), n, C, F16, w.data_ptr(), b.data_ptr(), F32,
                                       stats[0].data_ptr(), stats[1].data_ptr(), 1, 1, dw.data_ptr(), db.data_ptr(),
                                       ws.data_ptr(), ws.numel(), None, raw))

    Gc = (n + 127) // 128                      # records a convolution's epilogue leaves: one per 128-row tile
    recs = torch.zeros(3, C, Gc, device=dev)      # [field][channel][record]
    recs[0] = 128.0
    recs[1] = torch.randn(C, Gc, device=dev) * 0.1
    recs[2] = 128.0 + torch.rand(C, Gc, device=dev)

    def fwd_stats():
        _lib.check(L.spx_batchnorm_fwd_stats(x.data_ptr(), y.data_ptr(), n, C, F16, w.data_ptr(), b.data_ptr(), rm.data_ptr(),
                                             rv.data_ptr(), None, F32, 0.01, 1e-3, 1, stats[0].data_ptr(),
                                             stats[1].data_ptr(), recs.data_ptr(), Gc, None, raw))

    out = {"n": n, "C": C, "tensor_MB": round(n * C * 2 / 1e6, 1), "phases": os.environ.get("SPX_BN_PHASES", "7")}
    out["fwd_from_conv_records_us"] = round(timed(fwd_stats, s), 2)
    res = {}
    out["three_launches"] = {"fwd_us": round(timed(fwd, s), 2), "bwd_us": round(timed(bwd, s), 2)}
    out["ideal_us_at_6TBps"] = {"fwd": round(3 * n * C * 2 / 6e6, 2), "bwd": round(5 * n * C * 2 / 6e6, 2)}
    print(json.dumps(out), flush=True)
