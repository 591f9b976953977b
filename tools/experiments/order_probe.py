#!/usr/bin/env python
"""Row order vs forward time of igemm_v4 on the reference fixture scene (C = K = 64, fp16): given
(shuffled) order, mask-sorted, Morton-sorted, and coarse-cell-major / mask-minor orders, all with the
tables copied into tile order."""
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


def spread(v):
    v = v.astype(np.int64)
    r = np.zeros_like(v)
    for i in range(12):
        r |= ((v >> i) & 1) << (2 * i)
    return r


def main():
    dev = torch.device("cuda:0")
    kind = sys.argv[1] if len(sys.argv) > 1 else "fixture"
    C = 64
    idx, shape = bench.make_scene(kind, 100_000, 0)
    ind = torch.from_numpy(idx).to(dev)
    rb = ops.build_rulebook(ind, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)[0]
    n = idx.shape[0]
    f = torch.randn(n, C, device=dev).half()
    w = (torch.randn(C, 3, 3, 3, C, device=dev) * 0.1).half()
    mask = rb.mask_fwd.view(-1).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    y, x = idx[:, 2], idx[:, 3]
    orders = {"given": None, "mask": np.argsort(mask, kind="stable")}
    mort = (spread(y >> 1) << 1) | spread(x >> 1)
    orders["morton"] = np.argsort(mort, kind="stable")
    for cs in (5, 6, 7):            # cells of 32 / 64 / 128 voxels edge: cell-major, mask-minor
        cell = (spread(y >> cs) << 1) | spread(x >> cs)
        orders[f"cell{1 << cs}+mask"] = np.lexsort((mask, cell))
    L = _lib.load()
    res = {"scene": kind, "n": n}
    ref = ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, n, 13)
    for name, o in orders.items():
        if o is None:
            fn = lambda i: ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, n, 13)
        else:
            order = torch.from_numpy(o.astype(np.int32)).to(dev)
            pair_t, mask_t = torch.empty_like(rb.pair_fwd), torch.empty_like(rb.mask_fwd)
            _lib.check(L.spx_permute_tables(rb.pair_fwd.data_ptr(), rb.mask_fwd.data_ptr(), order.data_ptr(), 27, n, 1,
                                            pair_t.data_ptr(), mask_t.data_ptr(), ops._stream(f)))
            fn = lambda i, p=pair_t, m=mask_t, o_=order: ops.igemm_fwd(f, w, p, m, o_, n, 13, tile_order=True)
            assert torch.equal(fn(0), ref)
            mk = mask[o]
            m128 = mk[:(n // 128) * 128].reshape(-1, 128)
            steps = np.mean([bin(int(np.bitwise_or.reduce(r))).count("1") for r in m128])
            res[name + "_steps"] = round(float(steps), 1)
        res[name + "_us"] = round(1e3 * bench.event_time_ms(fn, span=8), 2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
