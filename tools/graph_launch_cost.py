"""Round 6: what a hipGraphLaunch costs the HOST as a function of the node count.  Graphs of n tiny kernels
(one 64-element add each, one stream / two parallel branches), replayed with an EMPTY queue (synchronize before every launch):
host time of the replay call and GPU time of the graph.  GPU box only."""
import time

import torch

dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)
y = torch.zeros(64, device=dev)


def capture(n, branches=1):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            if branches == 3:                       # a long chain with ONE short side branch at its head (2 nodes)
                s2.wait_stream(s)
                with torch.cuda.stream(s2):
                    y.add_(1.0)
                    y.add_(1.0)
                for _ in range(n - 2):
                    x.add_(1.0)
                s.wait_stream(s2)
            elif branches == 2:
                s2.wait_stream(s)
                with torch.cuda.stream(s2):
                    for _ in range(n // 2):
                        y.add_(1.0)
                for _ in range(n - n // 2):
                    x.add_(1.0)
                s.wait_stream(s2)
            else:
                for _ in range(n):
                    x.add_(1.0)
    return g


print("| nodes | branches | host time of replay(), empty queue (median of 9) | per node | GPU time of the graph | host time with 8 replays queued behind each other |")
print("|---|---|---|---|---|---|")
for n, br in ((1, 1), (10, 1), (100, 1), (500, 1), (1000, 1), (1000, 2), (2000, 2), (1000, 3)):
    g = capture(n, br)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    host, gpu = [], []
    for _ in range(9):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        g.replay()
        t1 = time.perf_counter()
        e1.record()
        torch.cuda.synchronize()
        host.append(t1 - t0)
        gpu.append(e0.elapsed_time(e1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        g.replay()
    tq = (time.perf_counter() - t0) / 8
    torch.cuda.synchronize()
    h = sorted(host)[4]
    print("| %d | %d | %.3f ms | %.2f us | %.3f ms | %.3f ms per replay |" % (n, br, h * 1e3, h / n * 1e6, sorted(gpu)[4], tq * 1e3))

# two single-chain graphs replayed CONCURRENTLY on two streams (what one two-branch graph expresses)
z = torch.zeros(64, device=dev)


def chain(buf, n, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        with torch.cuda.graph(g, stream=stream):
            for _ in range(n):
                buf.add_(1.0)
    return g


sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for n in (500, 1000):
    ga, gb = chain(x, n, sa), chain(z, n, sb)
    host, gpu = [], []
    for it in range(12):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cur = torch.cuda.current_stream()
        t0 = time.perf_counter()
        e0.record()
        sa.wait_stream(cur)
        sb.wait_stream(cur)
        with torch.cuda.stream(sa):
            ga.replay()
        with torch.cuda.stream(sb):
            gb.replay()
        cur.wait_stream(sa)
        cur.wait_stream(sb)
        t1 = time.perf_counter()
        e1.record()
        torch.cuda.synchronize()
        if it >= 3:
            host.append(t1 - t0)
            gpu.append(e0.elapsed_time(e1))
    print("| 2 x %d | two single-chain graphs on two streams | %.3f ms | %.2f us | %.3f ms | - |" % (n, sorted(host)[4] * 1e3, sorted(host)[4] / (2 * n) * 1e6, sorted(gpu)[4]))
