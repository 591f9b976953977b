"""Does a captured hipGraph with two independent branches run them concurrently on this stack?

A chain of small latency-bound kernels (the shape of a rulebook build) next to a chain of medium streaming kernels (the
shape of the convolution / normalisation chain), captured (a) on one stream, (b) forked onto a side stream and joined.
    python tools/probes/graph_branch_probe.py
"""
import json
import time

import torch


def main():
    dev = torch.device("cuda:0")
    small = [torch.zeros(4096, device=dev) for _ in range(4)]
    big_a = torch.zeros(8 << 20, device=dev, dtype=torch.float16)      # 16 MB
    big_b = torch.empty_like(big_a)
    n_small, n_big = 60, 30

    def small_chain():
        for i in range(n_small):
            small[i % 4].add_(1.0)

    def big_chain():
        for _ in range(n_big):
            big_b.copy_(big_a)
            big_a.add_(1)

    def serial():
        small_chain()
        big_chain()

    side = torch.cuda.Stream()

    def forked():
        main_s = torch.cuda.current_stream()
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            small_chain()
        big_chain()
        main_s.wait_stream(side)

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

    def timeit(run, iters=50):
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(iters):
            run()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / iters * 1e6

    res = {}
    res["eager_small_only_us"] = timeit(small_chain)
    res["eager_big_only_us"] = timeit(big_chain)
    res["eager_serial_us"] = timeit(serial)
    res["eager_forked_us"] = timeit(forked)
    for name, fn in (("small_only", small_chain), ("big_only", big_chain), ("serial", serial), ("forked", forked)):
        g = capture(fn)
        res[f"graph_{name}_us"] = timeit(g.replay)
    # (c) two LINEAR graphs replayed on two streams at once (no dependency between them)
    g_small, g_big = capture(small_chain), capture(big_chain)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def two_graphs():
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur)
        sb.wait_stream(cur)
        with torch.cuda.stream(sa):
            g_small.replay()
        with torch.cuda.stream(sb):
            g_big.replay()
        cur.wait_stream(sa)
        cur.wait_stream(sb)

    res["two_linear_graphs_two_streams_us"] = timeit(two_graphs)

    # (d) the same with a dependency in the middle through an EXTERNAL event (event-record / event-wait graph nodes instead
    # of a fork): the big chain's second half waits for the small chain's first half
    try:
        ev = torch.cuda.Event(external=True)

        def small_with_record():
            for i in range(n_small):
                small[i % 4].add_(1.0)
                if i == n_small // 2:
                    ev.record()

        def big_with_wait():
            for i in range(n_big):
                if i == n_big // 2:
                    torch.cuda.current_stream().wait_event(ev)
                big_b.copy_(big_a)
                big_a.add_(1)

        ev.record()
        torch.cuda.synchronize()
        g_s2, g_b2 = capture(small_with_record), capture(big_with_wait)

        def two_graphs_ev():
            cur = torch.cuda.current_stream()
            sa.wait_stream(cur)
            sb.wait_stream(cur)
            with torch.cuda.stream(sa):
                g_s2.replay()
            with torch.cuda.stream(sb):
                g_b2.replay()
            cur.wait_stream(sa)
            cur.wait_stream(sb)

        res["two_linear_graphs_external_event_us"] = timeit(two_graphs_ev)
    except Exception as e:      # noqa: BLE001
        res["two_linear_graphs_external_event_us"] = "unsupported: %r" % (e,)

    # (e) launch floor of a linear graph: 240 tiny kernels
    def tiny():
        for i in range(240):
            small[i % 4].add_(1.0)
    res["graph_240_tiny_us"] = timeit(capture(tiny).replay)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
