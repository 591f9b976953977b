#!/usr/bin/env python
"""Host-side (Python / ctypes / torch dispatch) cost of one backbone step on a single scene, where
the GPU is far from busy: cProfile of tools/netbench.py's cfg 4 network, batch 1."""
import cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import netbench
import spconv_amd.pytorch as spconv
from spconv_amd.utils import synthetic

dev = torch.device("cuda:0")
idx = torch.from_numpy(synthetic.lidar_like_scene(netbench.SHAPE, 100_000, 1, seed=0)).to(dev)
n = idx.shape[0]
net = netbench.backbone(4).to(dev).half()
f4 = torch.randn(n, 4, device=dev).half()

def step():
    net.zero_grad(set_to_none=True)
    out = net(spconv.SparseConvTensor(f4, idx, netbench.SHAPE, 1))
    out.features.float().square().mean().backward()

def infer():
    with torch.no_grad():
        net(spconv.SparseConvTensor(f4, idx, netbench.SHAPE, 1))

for fn, name in ((step, "train step"), (infer, "inference")):
    if name == "inference":
        net.eval()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    t_host = (time.perf_counter() - t0) / 20 * 1e3        # enqueue time only
    torch.cuda.synchronize()
    t_total = (time.perf_counter() - t0) / 20 * 1e3
    print(f"== {name}: host enqueue {t_host:.2f} ms, with GPU drain {t_total:.2f} ms per iteration")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        fn()
    pr.disable()
    torch.cuda.synchronize()
    sio = io.StringIO()
    pstats.Stats(pr, stream=sio).sort_stats("tottime").print_stats(18)
    print("\n".join(l[:150] for l in sio.getvalue().splitlines()[4:34]))
