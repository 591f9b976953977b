"""Two LINEAR graphs replayed on two streams at once: when does the second one start?  (rocprofv3 --kernel-trace around
`run`, then `analyse <csv>`: per queue, first start / last end of the final pair.)
    python tools/probes/graph_pair_probe.py run <n_small> <n_big> <small_first 0|1>"""
import csv
import sys


def run():
    import torch
    n_small, n_big, small_first = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dev = torch.device("cuda:0")
    small = [torch.zeros(4096, device=dev) for _ in range(4)]
    big_a = torch.zeros(8 << 20, device=dev, dtype=torch.float16)
    big_b = torch.empty_like(big_a)

    def small_chain():
        for i in range(n_small):
            small[i % 4].add_(1.0)

    def big_chain():
        for _ in range(n_big):
            big_b.mul_(1.0001)

    def capture(fn):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return g
    gs, gb = capture(small_chain), capture(big_chain)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(4):
        torch.cuda.synchronize()
        order = [(sa, gs), (sb, gb)] if small_first else [(sb, gb), (sa, gs)]
        for st, g in order:
            with torch.cuda.stream(st):
                g.replay()
    torch.cuda.synchronize()


def analyse():
    rows = []
    with open(sys.argv[2]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id")))
    rows.sort()
    n = int(sys.argv[3])                  # kernels of the last pair
    rows = rows[-n:]
    t0 = rows[0][0]
    qs = {}
    for s, e, name, q in rows:
        d = qs.setdefault(q, [s, e, 0, name[:40]])
        d[1] = max(d[1], e)
        d[2] += 1
    for q, (s, e, c, name) in qs.items():
        print(f"queue {q}: {c} kernels, first start {(s - t0) / 1e3:.1f} us, last end {(e - t0) / 1e3:.1f} us   {name}")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else analyse()
