"""Epilogue cost of the default GEMM at the fc1 / proj shapes (GPU box only; measurement helper)."""
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
for (N, K) in [(4608, 1152), (1152, 1152), (1152, 4608)]:
    x = torch.randn(1, M, K, generator=g).half().to(dev)
    W = (torch.randn(N, K, generator=g) * 0.03).half().to(dev)
    qa = ops.rowquant(x)
    d, z = ops.weight_minmax(W, 8)
    pw = ops.pack_weight(W, d, z, 8)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    res = torch.randn(M, N, generator=g).half().to(dev)
    gate = torch.randn(1, N, generator=g).float().to(dev)
    for name, kw in [("none", {}), ("gelu", dict(epilogue=ops.EPI_GELU)), ("resid", dict(epilogue=ops.EPI_RESID, resid=res)),
                     ("gate_resid", dict(epilogue=ops.EPI_GATE_RESID, resid=res, gate=gate, rows_per_gate=M))]:
        t = timeit(lambda: ops.gemm_i8(qa, pw, out=out, **kw), iters=30)
        print("N%d K%d %-10s %.1f us" % (N, K, name, t * 1e6), flush=True)
