"""Fixed (k-independent) cost of the ring GEMM: K = 128 launches at the STDiT output shapes vs a plain
fill of the same output buffer.  GPU box only; measurement helper."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops
from tools.bench_kernels import timeit
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab"))
import lab  # noqa: E402  (tools/lab: retired variants / probes live outside the product library)

dev = torch.device("cuda:0")
M = 16384
g = torch.Generator().manual_seed(0)
for N in (1152, 3456, 4608):
    for K in (128, 256, 1152):
        x = torch.randn(1, M, K, generator=g).half().to(dev)
        W = (torch.randn(N, K, generator=g) * 0.03).half().to(dev)
        qa = ops.rowquant(x)
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        d, z = ops.weight_minmax(W, 8)
        pw = ops.pack_weight(W, d, z, 8)
        for v in (10, 11):
            t = timeit(lambda: lab.gemm_i8(qa, pw, out=out, variant=v), iters=50)
            print("N%d K%d v%d: %.1f us" % (N, K, v, t * 1e6), flush=True)
    t = timeit(lambda: out.fill_(1.0), iters=50)
    print("N%d fill fp16 [M,N]: %.1f us  (%.2f TB/s)" % (N, t * 1e6, M * N * 2 / t / 1e12))
    t = timeit(lambda: out.copy_(out2) if False else None, iters=50)
    print("empty python loop: %.2f us" % (t * 1e6))
