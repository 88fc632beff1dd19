"""Temporal attention + proj quantizer: two kernels vs the fused one (STDiT 16x512x512 shape).  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops
from tools.bench_kernels import timeit

dev = torch.device("cuda:0")
T, S, H, D = 16, 1024, 16, 72
Cc = H * D
g = torch.Generator().manual_seed(0)
qkv = torch.randn(T * S, 3 * Cc, generator=g).half().to(dev)
o = torch.empty((T * S, Cc), dtype=torch.float16, device=dev)
junk = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)     # 256 MB: evicts L2 / MALL between runs


def two():
    ops.attn_temporal(qkv, qkv[:, Cc:], qkv[:, 2 * Cc:], o, 1, T, S, H, D, 3 * Cc, Cc)
    return ops.rowquant(o.view(1, T * S, Cc))


def fused():
    return ops.attn_temporal_rowquant(qkv, qkv[:, Cc:], qkv[:, 2 * Cc:], 1, T, S, H, D, 3 * Cc)


for name, fn in (("attention", lambda: ops.attn_temporal(qkv, qkv[:, Cc:], qkv[:, 2 * Cc:], o, 1, T, S, H, D, 3 * Cc, Cc)),
                 ("attention + rowquant", two), ("fused", fused)):
    t = timeit(fn, iters=50)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cold = []
    for _ in range(5):
        junk.fill_(1.0)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        cold.append(e0.elapsed_time(e1) * 1e3)
    print("%-22s warm %.1f us   cold (after a 256 MB fill) %.1f us" % (name, t * 1e6, min(cold)), flush=True)
