"""Two independent GEMM chains on two streams (as the cond / uncond branches of a step run): total time per pair of
launches for the full-CU kernel (variant 11) vs the half-CU kernels (20: one tile per workgroup, 21: persistent,
one workgroup per CU per stream).  GPU box only."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab"))
import lab  # noqa: E402  (tools/lab: retired variants / probes live outside the product library)

dev = torch.device("cuda:0")
M = 16384
g = torch.Generator().manual_seed(0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for (N, K) in [(1152, 1152), (3456, 1152), (4608, 1152), (1152, 4608)]:
    x = torch.randn(1, M, K, generator=g).half().to(dev)
    W = (torch.randn(N, K, generator=g) * 0.03).half().to(dev)
    qa = ops.rowquant(x)
    d, z = ops.weight_minmax(W, 8)
    pw = ops.pack_weight(W, d, z, 8)
    o1 = torch.empty((M, N), dtype=torch.float16, device=dev)
    o2 = torch.empty((M, N), dtype=torch.float16, device=dev)
    for v in (11, 20, 21):
        def run(n):
            for _ in range(n):
                with torch.cuda.stream(s1):
                    lab.gemm_i8(qa, pw, out=o1, variant=v)
                with torch.cuda.stream(s2):
                    lab.gemm_i8(qa, pw, out=o2, variant=v)
        try:
            run(3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(20)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / 20
            print("N%d K%d v%d: %.1f us per PAIR of launches on two streams (%.0f TOPS)" % (N, K, v, t * 1e6, 4.0 * M * N * K / t / 1e12), flush=True)
        except Exception as e:  # noqa
            print("N%d K%d v%d: %s" % (N, K, v, e), flush=True)
