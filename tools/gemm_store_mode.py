"""GEMM output store policy experiment (VQ_GEMM_STORE = 0 plain / 1 nontemporal / 2 write-through sc0 sc1):
kernel time at the STDiT shapes.  The env var is read once per process, so run once per mode.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops
from tools.bench_kernels import timeit

dev = torch.device("cuda:0")
M = 16384
g = torch.Generator().manual_seed(0)
res = []
for N, K, epi in ((1152, 1152, "none"), (1152, 1152, "resid"), (3456, 1152, "none"), (4608, 1152, "gelu"), (1152, 4608, "resid")):
    x = torch.randn(1, M, K, generator=g).half().to(dev)
    W = (torch.randn(N, K, generator=g) * 0.03).half().to(dev)
    qa = ops.rowquant(x)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    r = torch.randn(M, N, generator=g).half().to(dev)
    d, z = ops.weight_minmax(W, 8)
    pw = ops.pack_weight(W, d, z, 8)
    kw = {}
    if epi == "resid":
        kw = dict(epilogue=ops.EPI_RESID, resid=r)
    elif epi == "gelu":
        kw = dict(epilogue=ops.EPI_GELU)
    t = timeit(lambda: ops.gemm_i8(qa, pw, out=out, **kw), iters=100)
    res.append("N%d K%d %s %.1f" % (N, K, epi, t * 1e6))
print("store_mode", os.environ.get("VQ_GEMM_STORE", "0"), " | ".join(res), flush=True)
