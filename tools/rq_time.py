"""Per-token quantizer timings at the STDiT shapes (GPU box only): plain / LN + modulate / smoothed, C = 1152 and 4608."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops
dev = torch.device("cuda:0")
M = 16384
g = torch.Generator().manual_seed(0)


def timeit(fn, n=24, reps=12):
    """GPU time per call: n calls captured into one graph (the Python wrappers cost 10-20 us of host time per call - more than
    some of these kernels - so back-to-back eager launches measure the host), replayed reps times after a warm replay phase"""
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st):
            for _ in range(n):
                fn()
    for _ in range(20):
        gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


x = torch.randn(1, M, 1152, generator=g).half().to(dev)
x4 = torch.randn(1, M, 4608, generator=g).half().to(dev)
sh = (torch.randn(1, 1152, generator=g) * 0.3).float().to(dev)
sc = (torch.randn(1, 1152, generator=g) * 0.3).float().to(dev)
sm = [torch.exp(torch.randn(1152, generator=g) * 0.5).float().to(dev) for _ in range(3)]
sm4 = torch.exp(torch.randn(4608, generator=g) * 0.5).float().to(dev)
# a second, different buffer alternated with the first so that the input is not L2 / MALL resident from the last call
xb = torch.randn(1, M, 1152, generator=g).half().to(dev)
big = [torch.randn(1, M, 1152, generator=g).half().to(dev) for _ in range(12)]
i = [0]


def rot():
    i[0] = (i[0] + 1) % len(big)
    return big[i[0]]


print("rowquant C=1152              %6.1f us (rotating inputs %6.1f)" % (timeit(lambda: ops.rowquant(x)), timeit(lambda: ops.rowquant(rot()))))
print("LN+mod+quant C=1152          %6.1f us (rotating inputs %6.1f)" % (timeit(lambda: ops.ln_modulate_rowquant(x, sh, sc)), timeit(lambda: ops.ln_modulate_rowquant(rot(), sh, sc))))
print("rowquant smooth C=1152       %6.1f us" % timeit(lambda: ops.rowquant(rot(), s=sm[0])))
print("LN+mod+1 smooth C=1152       %6.1f us" % timeit(lambda: ops.ln_modulate_rowquant(rot(), sh, sc, smooth=sm[:1])))
print("rowquant_multi 3 x C=1152    %6.1f us" % timeit(lambda: ops.rowquant_multi(rot(), sm)))
print("LN+mod+3 smooth C=1152       %6.1f us" % timeit(lambda: ops.ln_modulate_rowquant(rot(), sh, sc, smooth=sm)))
print("rowquant C=4608              %6.1f us" % timeit(lambda: ops.rowquant(x4)))
print("rowquant smooth C=4608       %6.1f us" % timeit(lambda: ops.rowquant(x4, s=sm4)))
print("GELU + rowquant C=4608       %6.1f us" % timeit(lambda: ops.gelu_rowquant(x4)))
print("GELU + rowquant smooth 4608  %6.1f us" % timeit(lambda: ops.gelu_rowquant(x4, s=sm4)))
