#!/usr/bin/env python
"""Is a replay of the captured config-4 training step paced by the host?  Times `graph.replay()` (the call) and the replay
including its completion, for the forked graph (rulebook chain on a side branch) and the one-branch graph.
    python tools/graph_host_probe.py            (SPCONV_AMD_PREFETCH=0 for the one-branch form)"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from spconv_amd.pytorch.static import StaticTrainingStep, strided_layers  # noqa: E402
from spconv_amd.utils import nets  # noqa: E402
import spconv_amd.pytorch as spconv  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    bs, shape = 4, nets.SECOND_SHAPE
    net = nets.second_backbone(4).to(dev).half().train()
    idx, _ = bench.make_scene("lidar", 100_000, seed=0, batch=bs, shape=shape)
    ind = torch.from_numpy(idx).to(dev)
    f = torch.randn(idx.shape[0], 4, device=dev).half()
    seen = {}
    import copy
    probe = copy.deepcopy(net)
    hs = [m.register_forward_hook(lambda mod, a, out, k=k: seen.__setitem__(k, out.features.shape[0]))
          for k, m in strided_layers(probe).items()]
    with torch.no_grad():
        probe(spconv.SparseConvTensor(f, ind, shape, bs))
    bounds = {k: int(v * 1.1) + 1 for k, v in seen.items()}
    last = list(strided_layers(net).values())[-1]
    g = ((torch.rand((bounds[list(bounds)[-1]], last.out_channels), device=dev) - 0.5) * 0.2).half()
    step = StaticTrainingStep(net, int(idx.shape[0] * 1.05) + 1, 4, shape, bs, torch.float16, bounds=bounds, out_grad=g,
                              example=(f, ind))
    step.load(f, ind)
    for _ in range(10):
        step.graph.replay()
    torch.cuda.synchronize()
    call, total = [], []
    for _ in range(50):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step.graph.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        call.append(t1 - t0)
        total.append(t2 - t0)
    # back to back: 20 replays, one synchronisation
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step.graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    med = lambda v: sorted(v)[len(v) // 2]
    print(json.dumps({"prefetch": os.environ.get("SPCONV_AMD_PREFETCH", "auto"),
                      "replay_call_ms": round(med(call) * 1e3, 4), "replay_to_completion_ms": round(med(total) * 1e3, 4),
                      "twenty_back_to_back": {"calls_ms": round((t1 - t0) * 1e3, 3), "to_completion_ms": round((t2 - t0) * 1e3, 3),
                                              "per_step_ms": round((t2 - t0) * 1e3 / 20, 4)}}))


if __name__ == "__main__":
    main()
