"""Row-quantizer timing vs the persistent-grid cap VQ_RQ_GRID (GPU box only; run once per cap value)."""
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
x = torch.randn(1, M, 1152, generator=g).half().to(dev)
x4 = torch.randn(1, M, 4608, generator=g).half().to(dev)
sh = torch.randn(1, 1152, generator=g).float().to(dev)
t1 = timeit(lambda: ops.rowquant(x), iters=50)
t2 = timeit(lambda: ops.rowquant(x4), iters=50)
t3 = timeit(lambda: ops.ln_modulate_rowquant(x, sh, sh), iters=50)
print("VQ_RQ_GRID=%s  rowquant C1152 %.1f us (%.2f TB/s)  C4608 %.1f us (%.2f TB/s)  ln_mod_quant %.1f us (%.2f TB/s)" % (
    os.environ.get("VQ_RQ_GRID", "default"), t1 * 1e6, M * 1152 * 3 / t1 / 1e12, t2 * 1e6, M * 4608 * 3 / t2 / 1e12,
    t3 * 1e6, M * 1152 * 3 / t3 / 1e12))
t4 = timeit(lambda: ops.gelu_rowquant(x4), iters=50)
print("gelu_rowquant C4608 %.1f us (%.2f TB/s)" % (t4 * 1e6, M * 4608 * 3 / t4 / 1e12))
