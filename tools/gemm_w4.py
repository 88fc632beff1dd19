"""W4A8 vs W8A8 GEMM timing at the STDiT shapes (GPU box only; measurement helper)."""
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
for (N, K) in [(1152, 1152), (3456, 1152), (4608, 1152), (1152, 4608)]:
    x = torch.randn(1, M, K, generator=g).half().to(dev)
    W = (torch.randn(N, K, generator=g) * 0.03).half().to(dev)
    qa = ops.rowquant(x)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    for bits in (8, 4):
        d, z = ops.weight_minmax(W, bits)
        pw = ops.pack_weight(W, d, z, bits)
        for v in (11, 20, 21):
            t = timeit(lambda: lab.gemm_i8(qa, pw, out=out, variant=v), iters=30)
            print("N%d K%d W%d v%d: %.1f us  %.0f TOPS" % (N, K, bits, v, t * 1e6, 2.0 * M * N * K / t / 1e12), flush=True)
