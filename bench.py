#!/usr/bin/env python
"""Benchmark of the hot path: one 3x3x3 SubMConv3d, C=64 -> 64, fp16, ~100k active voxels in a
KITTI-shape grid (BASELINE.json configs[1]); metric = active voxels/s, forward + backward.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = forward + backward (dgrad + wgrad) of the layer over one resident scene batch with the
rulebook reused through `indice_key` (the reference shares SubM rulebooks the same way,
docs/USAGE.md:104-105); the rulebook build is timed separately and reported in `rulebook_ms`.
Inputs are resident in HBM before the timed region.  With N > 1 every rank owns its own scene
(weak scaling) and the step ends with one RCCL all-reduce of the weight gradient.

One JSON line is printed by rank 0; `roofline` is measured live with HIP events on the launch
stream for each kernel group, `cpu_baseline` times the CPU oracle (a port of the reference's
ConvAlgo.Native CPU path, oracle/) on a bounded sample at N = 1.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SHAPE = [40, 1280, 1600]       # z, y, x (KITTI-shape, SURVEY.md section 8d cfg 2)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--voxels", type=int, default=100_000)
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--scene", choices=["uniform", "lidar"], default="uniform")
    ap.add_argument("--dtype", choices=["f16", "bf16", "f32"], default="f16")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sort", action="store_true", help="mask_argsort the rulebook rows")
    ap.add_argument("--graph-steps", type=int, default=8,
                    help="steps captured per hipGraph at N = 1 (a replay boundary costs ~5 us; with N > 1 "
                         "the gradient all-reduce follows every step, so one step per replay)")
    return ap.parse_args()


def algorithmic_bytes(n, P, C, K, kv, s):
    """Compulsory bytes per call: features/outputs touched once, whole
    rulebook read once, weights once."""
    fwd = s * n * C + s * n * K + 4 * kv * n + s * kv * C * K
    dgrad = s * n * K + s * n * C + 4 * kv * n + s * kv * C * K
    wgrad = s * n * C + s * n * K + 4 * kv * n + 4 * kv * C * K
    return {"fwd": fwd, "dgrad": dgrad, "wgrad": wgrad}


def event_time_ms(fn, iters=30, warm=10):
    """Average device time of fn() between HIP events recorded on the current (launch) stream."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def pmc_traffic(group, args):
    """HBM bytes per launch of the dominant kernel group from the committed rocprofv3 PMC passes
    (FETCH_SIZE x 2 [gfx950 wide-read correction, MI355X_MICROARCH.md] + WRITE_SIZE, separate
    --pmc runs of THIS command; tools/pmc_traffic.py wrote profiles/traffic.json).  None when
    no measurement of the current configuration is on file."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        key = f"{args.scene}-{args.dtype}-c{args.channels}-n{args.voxels}"
        return t[key][group]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(idx, C, K, seed):
    """The oracle (port of the reference CPU path) on this host: rulebook once, then fwd+bwd.

    Two variants: faithful-pip = serial gather / scatter-add (the published
    CPU wheel has no OpenMP) + torch.mm on the thread pool; faithful-omp = rows gathered / scattered
    in parallel as a source build with -fopenmp would.  Thread settings: 16, and os.cpu_count() as
    BASELINE.md asks -- on a many-core host the small per-offset GEMMs oversubscribe badly there,
    so that setting gets one pass only when it is slow.  `value` is the FASTEST of all runs."""
    import oracle
    n = idx.shape[0]
    rng = np.random.default_rng(seed)
    f = torch.from_numpy(rng.uniform(-1, 1, (n, C)).astype(np.float32))
    w = torch.from_numpy(rng.uniform(-1, 1, (K, 3, 3, 3, C)).astype(np.float32))
    dout = torch.from_numpy(rng.uniform(-0.2, 0.2, (n, K)).astype(np.float32))
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    _, pair, num, _ = oracle.get_indice_pairs(idx, 1, SHAPE, [3] * 3, [1] * 3, [1] * 3, [1] * 3, subm=True)
    t_rule = time.perf_counter() - t0

    def one_pass(omp):
        t0 = time.perf_counter()
        oracle.indice_conv(f, w, pair, num, n, subm=True, omp=omp)
        oracle.indice_conv_backward(f, w, dout, pair, num, subm=True, omp=omp)
        return time.perf_counter() - t0

    results = {}
    slow_at_all_cores = False
    for variant, omp in (("faithful-pip", False), ("faithful-omp", True)):
        for threads in sorted({min(16, cores), cores}):
            if threads == cores and cores > 16 and slow_at_all_cores:
                continue                                    # already shown to oversubscribe: skip
            torch.set_num_threads(threads)
            oracle.set_omp_threads(threads)
            first = one_pass(omp)                           # warm-up, also sizes the budget
            times = []
            if first > 3.0:                                 # seconds per step: one pass is the sample
                times = [first]
                slow_at_all_cores = slow_at_all_cores or threads == cores
            else:
                budget = time.perf_counter() + 6.0
                while len(times) < 10 and (time.perf_counter() < budget or not times):
                    times.append(one_pass(omp))
            results[f"{variant}@{threads}"] = statistics.median(times)
    best = min(results, key=results.get)
    med = results[best]
    others = ", ".join(f"{k}: {v * 1e3:.0f} ms/step" for k, v in results.items())
    return {"value": n / med, "unit": "voxels/s", "cores": int(best.split("@")[1]), "kind": "port",
            "sample": f"fwd+bwd passes (1 warm-up, <= 10 timed, ~6 s budget per setting) of the same "
                      f"{n}-voxel scene, fp32, per-offset gather -> torch.mm -> scatter-add (BASELINE.md) on a "
                      f"{cores}-thread host; fastest = {best}; all settings: {others}; rulebook built once: "
                      f"{t_rule * 1e3:.1f} ms (single thread, std::unordered_map)",
            "variant": best.split("@")[0], "ms_per_step": med * 1e3, "rulebook_ms": t_rule * 1e3}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
    # BENCH_DIST_BACKEND=gloo BENCH_ONE_DEVICE=1: dry run of the N > 1 control flow on a single-GPU box
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_ONE_DEVICE", "0") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import spconv_amd.pytorch as spconv
    from spconv_amd.dist import GradBucket
    from spconv_amd.pytorch import ops
    from spconv_amd.utils import synthetic

    dtype = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[args.dtype]
    C = K = args.channels
    gen = synthetic.uniform_scene if args.scene == "uniform" else synthetic.lidar_like_scene
    idx_np = gen(SHAPE, args.voxels, 1, seed=rank)           # one scene per rank (weak scaling)
    n = idx_np.shape[0]
    indices = torch.from_numpy(idx_np).to(dev)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    feats = (torch.rand((n, C), generator=g) * 2 - 1).to(dev, dtype).requires_grad_(True)
    dout = ((torch.rand((n, K), generator=g) * 2 - 1) * 0.2).to(dev, dtype)
    torch.manual_seed(0)
    net = spconv.SubMConv3d(C, K, 3, bias=False, indice_key="bench").to(dev, dtype)
    net.train()

    # ---- rulebook: built once, reused by every step through indice_key ------------------
    def build():
        return ops.build_rulebook(indices, 1, SHAPE, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True,
                                  do_sort=args.sort)[0]
    rb = build()
    torch.cuda.synchronize()
    rule_ms = []
    for _ in range(5):
        t0 = time.perf_counter()
        build()
        torch.cuda.synchronize()
        rule_ms.append((time.perf_counter() - t0) * 1e3)
    x = spconv.SparseConvTensor(feats, indices, SHAPE, 1)
    x.indice_dict["bench"] = net._make_indice_data(rb, indices, SHAPE, SHAPE, net.algo)
    num = rb.num_per_loc.cpu().numpy()
    P = int(n + 2 * num[:13].sum())                            # pairs incl. centre
    bucket = GradBucket(net.parameters()) if world > 1 else None   # fp16 gradient, reduced in place

    def compute():
        net.weight.grad = None
        feats.grad = None
        y = net(x)
        y.features.backward(dout)

    launch = "eager"
    graph = None          # one step per replay
    graph_u = None        # U steps per replay (N = 1 only)
    U = max(1, args.graph_steps) if world == 1 else 1
    if not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    compute()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                compute()
            if U > 1:
                graph_u = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph_u):
                    for _ in range(U):
                        compute()
            launch = "hipgraph"
        except Exception as e:  # capture unsupported -> eager launches, same work
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); using eager launches",
                  file=sys.stderr)
            graph = graph_u = None
            torch.cuda.synchronize()
    if graph_u is None:
        U = 1

    # the process group comes up AFTER the capture, so that no RCCL helper thread can touch the
    # device while the stream is capturing
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def step():
        if graph is not None:
            graph.replay()
        else:
            compute()
        if bucket is not None:
            bucket.all_reduce(average=True)                    # one RCCL call per step

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    def run_steps(k):
        """Exactly k steps: whole U-step replays, then single steps."""
        if graph_u is not None:
            for _ in range(k // U):
                graph_u.replay()
            k = k % U
        for _ in range(k):
            step()

    run_steps(args.warmup)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    single_replay_ms = None
    if graph_u is not None:          # the same K steps with one step per replay, for reference
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        single_replay_ms = (time.perf_counter() - t1) / args.steps * 1e3
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([n], device=dev, dtype=torch.float64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        n_total = int(tot.item())
    else:
        n_total = n
    ms_per_step = elapsed / args.steps * 1e3
    value = n_total * args.steps / elapsed

    if rank == 0:
        # ---- per-kernel-group device time (HIP events on the launch stream) --------------
        w = net.weight.detach()
        fd = feats.detach()
        t_fwd = event_time_ms(lambda: ops.igemm_fwd(fd, w, rb.pair_fwd, rb.mask_fwd, rb.argsort_fwd, n, 13))
        t_dgrad = event_time_ms(lambda: ops.igemm_dgrad(dout, w, rb.pair_fwd, rb.mask_fwd,
                                                        rb.argsort_fwd, n, True))
        t_wgrad = event_time_ms(lambda: ops.igemm_wgrad(fd, dout, w.shape, rb.pair_native,
                                                        rb.num_per_loc, True, ops._plan_of(rb)))
        # the backward of a layer is ONE launch (igemm_bwd_kernel: dgrad tiles and wgrad ranges side
        # by side) plus the wgrad second stage; dgrad / wgrad alone are reported for reference
        t_bwd = event_time_ms(lambda: ops.igemm_bwd(fd, dout, w, rb.pair_fwd, rb.mask_fwd, rb.argsort_fwd,
                                                    rb.pair_native, rb.num_per_loc, True, ops._plan_of(rb)))
        t_eager = event_time_ms(compute, iters=20, warm=5)
        s = feats.element_size()
        ab = algorithmic_bytes(n, P, C, K, 27, s)
        ab["bwd"] = ab["dgrad"] + ab["wgrad"]
        groups = {"fwd": t_fwd, "bwd": t_bwd}
        alone = {"dgrad": t_dgrad, "wgrad": t_wgrad}
        kernels = {k: {"ms": round(v, 5), "algorithmic_MB": round(ab[k] / 1e6, 3),
                       "GBps": round(ab[k] / (v * 1e-3) / 1e9, 1)} for k, v in {**groups, **alone}.items()}
        dom = max(groups, key=groups.get)
        achieved = ab[dom] / (groups[dom] * 1e-3) / 1e9
        # stricter count for the fused backward launch: dout is read once for both gradients
        # (dout + feat + din + pair table + Native lists of P pairs + W + dW)
        strict = {"fwd": ab["fwd"],
                  "bwd": s * n * K + 2 * s * n * C + 4 * 27 * n + 8 * P + 2 * s * 27 * C * K}
        total_bytes = ab["fwd"] + ab["bwd"]
        result = {
            "metric": "active-voxels/sec fwd+bwd, 3x3x3 SubMConv3d C=64, ~100k voxels/scene",
            "value": value, "unit": "voxels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"SubMConv3d 3x3x3 C={C}->{K} {args.dtype}, {n} {args.scene}-random "
                                   f"voxels/scene in {SHAPE[2]}x{SHAPE[1]}x{SHAPE[0]} (BASELINE configs[1]), "
                                   f"1 scene per GPU, rulebook reused via indice_key",
                       "voxels_per_gpu": n, "pairs_per_voxel": round(P / n, 4), "launch": launch,
                       "steps_per_replay": U if launch == "hipgraph" else None,
                       "mask_sort": bool(args.sort), "parallelism": f"dp{world}"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": pmc_traffic(dom, args),
                         "kernel": {"fwd": "igemm_v4_kernel<64,2,f16,fwd>",
                                    "bwd": "igemm_bwd_kernel<64,2,f16> + wgrad_reduce2_kernel"}[dom],
                         "algorithmic_bytes": ab[dom], "ms": round(groups[dom], 5),
                         "shared_input_bytes": strict[dom],
                         "shared_input_frac": round(strict[dom] / (groups[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "kernels": kernels,
            "step_GBps_algorithmic": round(total_bytes / (ms_per_step * 1e-3) / 1e9, 1),
            "eager_device_ms_per_step": round(t_eager, 5),
            "ms_per_step_one_step_per_replay": None if single_replay_ms is None else round(single_replay_ms, 5),
            "rulebook_ms": round(statistics.median(rule_ms), 4),
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(idx_np, C, K, seed=1)
        print(json.dumps(result), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
