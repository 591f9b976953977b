#!/usr/bin/env python
"""Host-paced eager step (SubMConv3d forward + backward, BASELINE config 2) under a CPU affinity.

    python tools/eager_probe.py [cpulist | node:N | none]

Prints the device time per step of 3 windows of 200 eager steps, the CPU the thread ran on before / after, and the NUMA
node the GPU hangs off.  (bench.py's eager figure is bimodal from process to process: 75 vs 150-200 us.)"""
import glob
import json
import os
import sys


def cur_cpu():
    return int(open("/proc/self/stat").read().rsplit(")", 1)[1].split()[36])     # field 39: processor


def parse_cpus(s):
    out = []
    for part in s.strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out


sel = sys.argv[1] if len(sys.argv) > 1 else "none"
if sel.startswith("node:"):
    cpus = parse_cpus(open(f"/sys/devices/system/node/node{sel[5:]}/cpulist").read())
    os.sched_setaffinity(0, cpus)
elif sel != "none":
    os.sched_setaffinity(0, parse_cpus(sel))

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spconv_amd.pytorch as spconv  # noqa: E402
from spconv_amd.utils import synthetic  # noqa: E402

SHAPE = [40, 1280, 1600]
dev = torch.device("cuda:0")
props = torch.cuda.get_device_properties(0)
bdf = None
try:
    bdf = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
    gpu_node = open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip()
except Exception as e:  # noqa: BLE001
    gpu_node = f"unknown ({e})"
nodes = sorted(glob.glob("/sys/devices/system/node/node[0-9]*"))
n = 100_000
idx = torch.from_numpy(synthetic.uniform_scene(SHAPE, n, 1, seed=0)).to(dev)
net = spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="k").to(dev).half()
f = (torch.rand(n, 64, device=dev) * 2 - 1).half().requires_grad_(True)
x = spconv.SparseConvTensor(f, idx, SHAPE, 1)
dout = (torch.rand(n, 64, device=dev) * 2 - 1).half()
net(x).features.backward(dout)              # builds the rulebook (cached in x.indice_dict)


def step():
    net.weight.grad = None
    f.grad = None
    net(x).features.backward(dout)


cpu0 = cur_cpu()
runs = []
for _ in range(3):
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        step()
    b.record()
    torch.cuda.synchronize()
    runs.append(round(a.elapsed_time(b) / 200, 5))
print(json.dumps({"affinity": sel, "ms_per_step": runs, "cpu_before": cpu0, "cpu_after": cur_cpu(),
                  "gpu_pci": bdf, "gpu_numa_node": gpu_node, "numa_nodes": len(nodes),
                  "cpus_allowed": len(os.sched_getaffinity(0))}))
