"""Spatial / cross / PixArt attention timing (GPU box only).  VQ_ATTN_V1=1 selects the first-generation kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops
from tools.bench_kernels import timeit

dev = torch.device("cuda:0")
H, D = 16, 72
g = torch.Generator().manual_seed(0)
for name, n_seq, L in (("spatial 16 x 1024", 16, 1024), ("pixart 1 x 4096", 1, 4096)):
    M = n_seq * L
    qkv = torch.randn(M, 3 * 1152, generator=g).half().to(dev)
    o = torch.empty((M, 1152), dtype=torch.float16, device=dev)
    ld = 3456
    t = timeit(lambda: ops.attn_fwd(qkv, qkv[:, 1152:], qkv[:, 2304:], o, n_seq, L, L, H, D, L * ld, ld, L * ld, ld,
                                    L * 1152, 1152), iters=20)
    fl = 4.0 * n_seq * L * L * H * D
    print("%s: %.1f us  %.0f TFLOP/s (useful)" % (name, t * 1e6, fl / t / 1e12))
q = torch.randn(16384, 1152, generator=g).half().to(dev)
kv = torch.randn(120, 2304, generator=g).half().to(dev)
off = torch.tensor([0, 120], dtype=torch.int32, device=dev)
o = torch.empty_like(q)
t = timeit(lambda: ops.attn_fwd(q, kv, kv[:, 1152:], o, 1, 16384, 0, H, D, 16384 * 1152, 1152, 0, 2304, 16384 * 1152, 1152,
                                kv_off=off), iters=20)
print("cross 16384 x 120: %.1f us" % (t * 1e6))
for Lk in (120, 80):
    off = torch.tensor([0, Lk], dtype=torch.int32, device=dev)
    t = timeit(lambda: ops.attn_fwd(q, kv[:Lk], kv[:Lk, 1152:], o, 1, 16384, Lk, H, D, 16384 * 1152, 1152, 0, 2304, 16384 * 1152,
                                    1152, kv_off=off), iters=20)
    print("cross 16384 x %d, bound known (register kernel): %.1f us" % (Lk, t * 1e6))
qkv = torch.randn(16384, 3456, generator=g).half().to(dev)
o = torch.empty((16384, 1152), dtype=torch.float16, device=dev)
t = timeit(lambda: ops.attn_temporal(qkv, qkv[:, 1152:], qkv[:, 2304:], o, 1, 16, 1024, H, D, 3456, 1152), iters=20)
print("temporal 1024 x 16: %.1f us  (%.2f TB/s)" % (t * 1e6, 16384 * 1152 * 2 * 4 / t / 1e12))
