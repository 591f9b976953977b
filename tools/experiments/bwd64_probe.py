import json, os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from spconv_amd import _lib
from spconv_amd.pytorch import ops
from spconv_amd.utils import nets
L = _lib.load()
dev = torch.device("cuda:0")
idx, shape = bench.make_scene("lidar", 100_000, 0, batch=4, shape=nets.SECOND_SHAPE)
ind = torch.from_numpy(idx).to(dev)
t = lambda fn: round(1e3 * bench.event_time_ms(fn, span=4), 1)
for lvl in (2, 3, 4):
    rb, shape = ops.build_rulebook(ind, 4, shape, [3] * 3, [2] * 3, [1] * 3 if lvl < 4 else [0, 1, 1], [1] * 3, [0] * 3, False, out_order="sorted")
    ind = rb.out_indices
    if lvl < 3: continue
    n = ind.shape[0]
    sub = ops.build_rulebook(ind, 4, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)[0]
    C = 64
    f = torch.randn(n, C, device=dev).half(); d = torch.randn(n, C, device=dev).half()
    w = (torch.randn(C, 3, 3, 3, C, device=dev) * 0.1).half()
    plan = ops._plan_of(sub)
    r = dict(level=lvl, rows=n)
    r["fused_bwd"] = t(lambda i: ops.igemm_bwd(f, d, w, sub.pair_fwd, sub.mask_fwd, None, sub.pair_native, sub.num_per_loc, True, plan))
    r["wgrad"] = t(lambda i: ops.igemm_wgrad(f, d, w.shape, sub.pair_native, sub.num_per_loc, True, plan))
    for ws in (0, 1):
        L.spx_set_option(b"SPX_WS", ws)
        r[f"dgrad_ws{ws}"] = t(lambda i: ops.igemm_dgrad(d, w, sub.pair_fwd, sub.mask_fwd, None, n, True))
        r[f"fwd_ws{ws}"] = t(lambda i: ops.igemm_fwd(f, w, sub.pair_fwd, sub.mask_fwd, None, n, 13))
    L.spx_set_option(b"SPX_WS", -1)
    print(json.dumps(r), flush=True)
