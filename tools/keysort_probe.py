#!/usr/bin/env python
"""Entry sort of the static runners (spx_key_argsort, csrc/rowsort.hip) on BASELINE config 4's scenes: device time of the
sort, of the rank-map pass behind it and of torch's argsort on the same keys.
    python tools/keysort_probe.py [voxels per batch item]      (under rocprofv3 --kernel-trace --stats: per-kernel times)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from spconv_amd.pytorch import ops  # noqa: E402
from spconv_amd.utils import nets  # noqa: E402


def main():
    voxels = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    dev = torch.device("cuda:0")
    bs, shape = 4, nets.SECOND_SHAPE
    for kind in ("lidar", "uniform"):
        idx, _ = bench.make_scene(kind, voxels, seed=0, batch=bs, shape=shape)
        n = idx.shape[0]
        pad = int(n * 1.05) + 1
        ind = torch.full((pad, 4), -1, dtype=torch.int32, device=dev)
        ind[:n] = torch.from_numpy(idx).to(dev)
        f = torch.randn(pad, 4, device=dev).half()
        order, rows = ops.key_argsort(ind, bs, shape)
        key = ind[:, 0].long()
        for d, s in enumerate(shape):
            key = key * int(s) + ind[:, 1 + d].long()
        key[ind[:, 0] < 0] = torch.iinfo(torch.int64).max
        want = torch.argsort(key, stable=True)
        ok = bool(torch.equal(order.long(), want))
        top = (key[:n] >> 20).bincount()
        t_sort = bench.event_time_ms(lambda i: ops.key_argsort(ind, bs, shape), iters=40, span=4) * 1e3
        t_both = bench.event_time_ms(lambda i: ops.key_argsort(ind, bs, shape, rank_map=True), iters=40, span=4) * 1e3
        t_map = bench.event_time_ms(lambda i: ops.attach_rank_map(rows, bs, shape, check=False), iters=40, span=4) * 1e3
        t_gather = bench.event_time_ms(lambda i: f.index_select(0, order), iters=40, span=4) * 1e3
        t_torch = bench.event_time_ms(lambda i: torch.argsort(key), iters=20, span=2) * 1e3
        print(json.dumps({"scene": kind, "rows": pad, "live": n, "equals_stable_argsort": ok,
                          "key_argsort_us": round(t_sort, 1), "key_argsort_with_rank_map_us": round(t_both, 1),
                          "rank_map_from_sorted_us": round(t_map, 1),
                          "feature_gather_us": round(t_gather, 1), "torch_argsort_us": round(t_torch, 1),
                          "rows_in_largest_2^20_bucket": int(top.max()), "buckets_used": int((top > 0).sum())}))


if __name__ == "__main__":
    main()
