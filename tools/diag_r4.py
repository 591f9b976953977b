import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spconv_amd.pytorch import ops
from spconv_amd.utils import synthetic
dev = torch.device("cuda:0")
def say(*a):
    print(*a, flush=True)
shape = [40, 1280, 1600]
idx = torch.from_numpy(synthetic.uniform_scene(shape, 100_000, 1, 0)).to(dev)
say("scene ok")
for native in (False, True):
    rb, _ = ops.build_rulebook(idx, 1, shape, [3]*3, [1]*3, [1]*3, [1]*3, [0]*3, True, need_native=native)
    torch.cuda.synchronize(); say("rulebook ok native", native)
ops.rows_layout(rb); torch.cuda.synchronize(); say("layout ok", rb.layout[:4].tolist())
f = torch.randn(100_000, 64, device=dev).half(); w = torch.randn(64, 3, 3, 3, 64, device=dev).half()
o0 = ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, 13); torch.cuda.synchronize(); say("fwd rows ok")
p, m, a, to = ops.tables_of(rb, "fwd", 64); say("to", to)
o1 = ops.igemm_fwd(f, w, p, m, a, rb.n_out, 13, tile_order=to); torch.cuda.synchronize(); say("fwd layout ok", torch.equal(o0, o1))
d0 = ops.igemm_dgrad(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_in, True); torch.cuda.synchronize(); say("dgrad rows ok")
d1 = ops.igemm_dgrad(f, w, p, m, a, rb.n_in, True, tile_order=to); torch.cuda.synchronize(); say("dgrad layout ok", torch.equal(d0, d1))
plan = ops._plan_of(rb)
b0 = ops.igemm_bwd(f, f, w, rb.pair_fwd, rb.mask_fwd, None, rb.pair_native, rb.num_per_loc, True, plan); torch.cuda.synchronize(); say("bwd rows ok")
b1 = ops.igemm_bwd(f, f, w, p, m, a, rb.pair_native, rb.num_per_loc, True, plan, tile_order=to); torch.cuda.synchronize(); say("bwd layout ok", torch.equal(b0[0], b1[0]), torch.equal(b0[1], b1[1]))
