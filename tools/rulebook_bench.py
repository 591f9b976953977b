#!/usr/bin/env python
"""Device time of the rulebook builders (graph replay between HIP events: no host time in the figure).

    SubM k3 (inference tables only / with the Native lists), SparseConv k3 s2 p1 and k2 s2 in the
    static-shape form (spx_conv_rulebook_static: the same passes as the two-call form, nothing read
    back, so it can sit in a graph), second- vs third-generation passes (SPX_CONV_V = 2 / 3), the sorted-order
    build of the same layer (rank map) and the SubM layer behind it, hash build vs rank-map build.

Scenes: uniform 100 k (BASELINE config 2), LiDAR-like 100 k and 4 x 100 k (config 4 level 1), the
reference fixture.  One JSON line.   python tools/rulebook_bench.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from spconv_amd import _lib  # noqa: E402
from spconv_amd.pytorch import ops  # noqa: E402
from spconv_amd.utils import nets  # noqa: E402


def set_option(name, v):
    _lib.check(_lib.load().spx_set_option(name.encode(), int(v)))


def main():
    dev = torch.device("cuda:0")
    rows = []
    scenes = [("uniform", 100_000, 1, None), ("lidar", 100_000, 1, None), ("lidar", 100_000, 4, nets.SECOND_SHAPE),
              ("fixture", 0, 1, None)]
    for kind, n, bs, shape in scenes:
        idx, shape = bench.make_scene(kind, n, 0, batch=bs, shape=shape)
        ind = torch.from_numpy(idx).to(dev)
        N = ind.shape[0]
        row = dict(scene=kind, batch=bs, voxels=N)

        def subm(native, sort=False):
            return lambda i: ops.build_rulebook(ind, bs, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True,
                                                need_native=native, do_sort=sort)
        row["subm_tables_us"] = round(bench.event_time_ms(subm(False), iters=40, span=4) * 1e3, 1)
        # the probe pass without the LDS-staged occupancy bits (its fourth form), for comparison
        set_option("SPX_SUBM_PROBE", 4)
        row["subm_tables_probe4_us"] = round(bench.event_time_ms(subm(False), iters=40, span=4) * 1e3, 1)
        set_option("SPX_SUBM_PROBE", 5)
        row["subm_with_lists_us"] = round(bench.event_time_ms(subm(True), iters=40, span=4) * 1e3, 1)
        # what the layer modules build by default: + the rows layout (count -> scan -> scatter, nothing read back)
        row["subm_tables_layout_us"] = round(bench.event_time_ms(subm(False, "layout"), iters=40, span=4) * 1e3, 1)
        row["subm_with_lists_layout_us"] = round(bench.event_time_ms(subm(True, "layout"), iters=40, span=4) * 1e3, 1)
        rb = subm(True, "layout")(0)[0]
        if rb.layout is not None:
            row["layout_class"], row["rows_with_a_neighbour"] = rb.layout[:2].cpu().tolist()
        # round 6: the mask argsort alone (one launch per digit pass; 27 mask bits = three 9-bit passes, 32 = four 8-bit
        # ones) and the explicit sort of a rulebook (SPCONV_DO_SORT=1: argsort + table copies)
        row["mask_argsort_kv27_us"] = round(bench.event_time_ms(lambda i: ops.mask_argsort(rb.mask_fwd, 27), iters=40, span=4) * 1e3, 1)
        row["mask_argsort_32bit_us"] = round(bench.event_time_ms(lambda i: ops.mask_argsort(rb.mask_fwd), iters=40, span=4) * 1e3, 1)
        row["sort_rulebook_us"] = round(bench.event_time_ms(lambda i: ops.sort_rulebook(rb), iters=40, span=4) * 1e3, 1)
        # round 6: the same scene handed over in coordinate-key order (utils.sort_voxels_by_coordinate): level 1 from a
        # rank map built from the rows themselves (ops.attach_rank_map) instead of a hash table
        from spconv_amd.pytorch.utils import sort_voxels_by_coordinate
        ind_s = sort_voxels_by_coordinate(ind, shape, batch_size=bs, rank_map=False)[0]

        def keyed(sort):
            def fn(i):
                ops.attach_rank_map(ind_s, bs, shape, check=False)
                return ops.build_rulebook(ind_s, bs, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True,
                                          need_native=False, do_sort=sort)
            return fn
        if ops.attach_rank_map(ind_s, bs, shape, check=True):
            row["key_ordered_rank_map_us"] = round(bench.event_time_ms(
                lambda i: ops.attach_rank_map(ind_s, bs, shape, check=False), iters=40, span=4) * 1e3, 1)
            row["key_ordered_subm_tables_us"] = round(bench.event_time_ms(keyed(False), iters=40, span=4) * 1e3, 1)
            row["key_ordered_subm_tables_layout_us"] = round(bench.event_time_ms(keyed("layout"), iters=40, span=4) * 1e3, 1)
            ind_s._spx_rankmap = None
            hash_rb = ops.build_rulebook(ind_s, bs, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True, need_native=False)[0]
            row["key_ordered_equals_hash_build"] = bool(torch.equal(keyed(False)(0)[0].pair_fwd, hash_rb.pair_fwd))
        for name, k, s, p in (() if os.environ.get("RB_ONLY_SUBM") == "1" else (("conv_k3s2", 3, 2, 1), ("conv_k2s2", 2, 2, 0))):
            rb, _ = ops.build_rulebook(ind, bs, shape, [k] * 3, [s] * 3, [p] * 3, [1] * 3, [0] * 3, False)
            cap = rb.n_out + 1024
            row[name + "_n_out"] = rb.n_out
            ref = None
            for v in (2, 3):
                set_option("SPX_CONV_V", v)
                fn = lambda i: ops.build_rulebook(ind, bs, shape, [k] * 3, [s] * 3, [p] * 3, [1] * 3, [0] * 3, False,
                                                  need_native=False, static_num_out=cap)
                row[f"{name}_v{v}_us"] = round(bench.event_time_ms(fn, iters=40, span=4) * 1e3, 1)
                got = fn(0)[0]
                torch.cuda.synchronize()
                t = (got.out_indices, got.pair_fwd, got.pair_bwd, got.mask_fwd, got.mask_bwd)
                if ref is None:
                    ref = t
                else:
                    row[name + "_v3_equals_v2"] = all(torch.equal(a, b) for a, b in zip(ref, t))
            set_option("SPX_CONV_V", 3)
            # sorted-order level (rank map instead of the hash table) and the SubM layer behind it:
            # hash build of the same rows against the build over the level's rank map
            fs = lambda i: ops.build_rulebook(ind, bs, shape, [k] * 3, [s] * 3, [p] * 3, [1] * 3, [0] * 3, False,
                                              need_native=False, static_num_out=cap, out_order="sorted")
            got = fs(0)[0]
            if got.rankmap is not None:
                row[name + "_sorted_us"] = round(bench.event_time_ms(fs, iters=40, span=4) * 1e3, 1)
                lvl, oshape = got.out_indices, got.out_shape
                plain = lvl.clone()
                sub = lambda t: (lambda i: ops.build_rulebook(t, bs, oshape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3,
                                                              True, need_native=False))
                row[name + "_level_subm_hash_us"] = round(bench.event_time_ms(sub(plain), iters=40, span=4) * 1e3, 1)
                row[name + "_level_subm_ranked_us"] = round(bench.event_time_ms(sub(lvl), iters=40, span=4) * 1e3, 1)
                row[name + "_level_subm_equal"] = bool(torch.equal(sub(plain)(0)[0].pair_fwd, sub(lvl)(0)[0].pair_fwd))
        rows.append(row)
    print(json.dumps({"rulebook_device_us": rows}))


if __name__ == "__main__":
    main()
