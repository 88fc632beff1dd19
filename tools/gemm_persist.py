"""Variant 11 (one tile per workgroup) vs 14 (persistent, next tile's first stage prefetched under the epilogue) at the
multi-round STDiT shapes.  GPU box only."""
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
for N, K, epi in ((3456, 1152, ops.EPI_NONE), (4608, 1152, ops.EPI_GELU), (4608, 1152, ops.EPI_NONE), (2304, 1152, ops.EPI_NONE)):
    x = torch.randn(1, M, K, generator=g).half().to(dev)
    W = (torch.randn(N, K, generator=g) * 0.03).half().to(dev)
    b = torch.randn(N, generator=g).to(dev)
    qa = ops.rowquant(x)
    d, z = ops.weight_minmax(W, 8)
    pw = ops.pack_weight(W, d, z, 8)
    outs = {}
    for v in (11, 14):
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        lab.gemm_i8(qa, pw, bias=b, out=out, epilogue=epi, variant=v)
        t = timeit(lambda: lab.gemm_i8(qa, pw, bias=b, out=out, epilogue=epi, variant=v), iters=100)
        outs[v] = out
        print("N%d K%d epi%d v%d: %.1f us" % (N, K, epi, v, t * 1e6), flush=True)
    print("   bit-identical:", bool(torch.equal(outs[11], outs[14])), flush=True)
