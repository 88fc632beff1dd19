"""Ablation timings of the full-line ring GEMM (variant 11): which of DMA / MFMA / barrier / fragment
reads the main loop is waiting on.  GPU box only; results of the ablated launches are wrong by design."""
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
names = {11: "full", 101: "no DMA", 102: "no MFMA", 103: "no DMA, no MFMA", 104: "no barrier", 105: "no DMA no barrier",
         108: "no frag reads", 109: "no DMA no reads (MFMA + barrier)", 110: "no MFMA no reads (DMA + barrier)",
         112: "no barrier no reads", 113: "MFMA only"}
for (N, K) in [(1152, 4608), (1152, 1152)]:
    x = torch.randn(1, M, K, generator=g).half().to(dev)
    W = (torch.randn(N, K, generator=g) * 0.03).half().to(dev)
    qa = ops.rowquant(x)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    d, z = ops.weight_minmax(W, 8)
    pw = ops.pack_weight(W, d, z, 8)
    for v in (11, 101, 102, 103, 104, 105, 108, 109, 110, 112, 113):
        t = timeit(lambda: lab.gemm_i8(qa, pw, out=out, variant=v), iters=30)
        print("N%d K%d %-36s %.1f us" % (N, K, names[v], t * 1e6), flush=True)
