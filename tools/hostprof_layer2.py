#!/usr/bin/env python
"""Where the HOST time of one eager layer step goes (SubMConv3d 64 -> 64, 100 k uniform voxels, rulebook cached:
BASELINE config 2 without a graph): cProfile by internal time over 300 forward + backward steps."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spconv_amd.pytorch as spconv  # noqa: E402
from spconv_amd.utils import synthetic  # noqa: E402

if os.environ.get("HOSTPROF_SINGLE_THREAD_AUTOGRAD") == "1":
    torch.autograd.set_multithreading_enabled(False)       # backward on the calling thread: no hop to the device thread
if os.environ.get("HOSTPROF_CPUS"):
    os.sched_setaffinity(0, [int(c) for c in os.environ["HOSTPROF_CPUS"].split(",")])
dev = torch.device("cuda:0")
shape = [40, 1280, 1600]
idx = torch.from_numpy(synthetic.uniform_scene(shape, 100_000, 1, seed=0)).to(dev)
n = idx.shape[0]
net = spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="k").to(dev).half()
f = torch.randn(n, 64, device=dev).half().requires_grad_(True)
dout = torch.randn(n, 64, device=dev).half() * 0.1
with torch.no_grad():
    y0 = net(spconv.SparseConvTensor(f.detach(), idx, shape, 1))
x = spconv.SparseConvTensor(f, idx, shape, 1, indice_dict=y0.indice_dict)


def step():
    net.weight.grad = None
    f.grad = None
    net(x).features.backward(dout)


for _ in range(50):
    step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(300):
        step()
    th = (time.perf_counter() - t0) / 300 * 1e6
    torch.cuda.synchronize()
    print("host us per step %.1f, drained %.1f" % (th, (time.perf_counter() - t0) / 300 * 1e6))
if os.environ.get("HOSTPROF_TIMING_ONLY") == "1":
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
sio = io.StringIO()
pstats.Stats(pr, stream=sio).sort_stats("tottime").print_stats(40)
print("\n".join(l[:150] for l in sio.getvalue().splitlines()[4:52]))
