"""rocprofv3 --kernel-trace of a few pipelined inference passes (config-4 network): do the rulebook graph and the
convolution graph overlap on the device?   python tools/experiments/pipe_trace.py run | analyse <csv>"""
import sys, csv
if sys.argv[1] == "run":
    import numpy as np, torch
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    import bench
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch.static import RulebookPipeline, strided_layers
    from spconv_amd.utils import nets
    dev = torch.device("cuda:0")
    net = nets.second_backbone(4).to(dev).half().eval()
    bs, shape = 4, nets.SECOND_SHAPE
    data = []
    for si in range(3):
        idx_np, shape = bench.make_scene("lidar", 100_000, seed=si, batch=bs, shape=nets.SECOND_SHAPE)
        data.append((torch.from_numpy(idx_np).to(dev), torch.randn(idx_np.shape[0], 4, device=dev).half()))
    seen = {}
    hooks = [m.register_forward_hook(lambda mod, a, out, k=k: seen.__setitem__(k, max(seen.get(k, 0), out.features.shape[0])))
             for k, m in strided_layers(net).items()]
    with torch.no_grad():
        for ind, f in data:
            net(spconv.SparseConvTensor(f, ind, shape, bs))
    for h in hooks:
        h.remove()
    bounds = {k: int(v * 1.1) + 1 for k, v in seen.items()}
    n_max = max(d[0].shape[0] for d in data)
    pipe = RulebookPipeline(net, int(n_max * 1.05) + 1, 4, shape, bs, torch.float16, bounds=bounds)
    pipe.submit(data[0][1], data[0][0])
    import time
    torch.cuda.synchronize()
    t = time.perf_counter()
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    for k in range(N):
        ind, f = data[(k + 1) % 3]
        pipe.submit(f, ind)
        pipe.run()
    torch.cuda.synchronize()
    print("ms per step", (time.perf_counter() - t) / N * 1e3)
    # two independent graphs at once, both launch orders
    g_r, g_c = pipe.slots[0].graph_r, pipe.slots[1].graph_c
    cur, rb = torch.cuda.current_stream(), pipe.rb_stream
    def both(first_r, gap=0.0):
        rb.wait_stream(cur)
        if first_r:
            with torch.cuda.stream(rb):
                g_r.replay()
            if gap:
                time.sleep(gap)
            g_c.replay()
        else:
            g_c.replay()
            if gap:
                time.sleep(gap)
            with torch.cuda.stream(rb):
                g_r.replay()
        cur.wait_stream(rb)
    for first_r in (True, False):
        for gap in (0.0, 0.0001):
            for _ in range(3):
                both(first_r, gap)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(20):
                both(first_r, gap)
            torch.cuda.synchronize()
            print("R first" if first_r else "C first", "gap", gap, "us per pair", round((time.perf_counter() - t) / 20 * 1e6))
    # host time of the replay calls alone
    sl = pipe.slots[0]
    for name, g in (("R", sl.graph_r), ("C", sl.graph_c)):
        torch.cuda.synchronize()
        hs = []
        for _ in range(5):
            t = time.perf_counter()
            g.replay()
            hs.append((time.perf_counter() - t) * 1e6)
            torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            g.replay()
        h = (time.perf_counter() - t) / 20 * 1e6
        torch.cuda.synchronize()
        d = (time.perf_counter() - t) / 20 * 1e6
        print(name, "host us per replay (idle queue)", [round(x) for x in hs], "back to back host", round(h), "device", round(d))
else:
    rows = []
    with open(sys.argv[2]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50], r.get("Queue_Id"), r.get("Stream_Id")))
    rows.sort()
    rows = rows[-1500:]
    qs = {}
    for s, e, n, q, st in rows:
        qs.setdefault((q, st), []).append((s, e, n))
    print({k: len(v) for k, v in qs.items()})
    t0 = rows[0][0]
    for s, e, n, q, st in rows[-260:]:
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.2f} q{q} s{st} {n}")
