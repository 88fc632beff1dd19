"""128 x 288 tile (variant 16) against the 256 x 288 tile (variant 11) of the ring GEMM on the launches that fill at most
half the chip with 256-row tiles: PixArt-Sigma 1024^2 (M = 8192 = uncond | cond x 4096 tokens) N = 1152 Linears, the
prompt K/V of one block (M = 300 / 120, N = 2304) and the batched prompt K/V of all 28 STDiT blocks.  Back to back, 100
launches after 30 of warm-up.  GPU box only.   python tools/gemm_half_tiles.py [--w4]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops

dev = torch.device("cuda")
w_bits = 4 if "--w4" in sys.argv else 8
g = torch.Generator().manual_seed(0)
SHAPES = [(8192, 1152, 1152, ops.EPI_NONE, "cross-q"), (8192, 1152, 1152, ops.EPI_GATE_RESID, "proj+gate"),
          (8192, 1152, 1152, ops.EPI_RESID, "cross-proj"), (8192, 1152, 4608, ops.EPI_GATE_RESID, "fc2+gate"),
          (8192, 3456, 1152, ops.EPI_NONE, "qkv (384 tiles)"), (600, 2304, 1152, ops.EPI_NONE, "kv Lp 2x300"),
          (4096, 1152, 1152, ops.EPI_NONE, "M 4096"), (2048, 1152, 1152, ops.EPI_NONE, "M 2048")]


def timed(fn, n=100, warm=30):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for M, N, K, epi, name in SHAPES:
    x = (torch.randn(1, M, K, generator=g) * 1.5).half().to(dev)
    W = (torch.randn(N, K, generator=g) * 0.04).half().to(dev)
    qa = ops.rowquant(x)
    d, z = ops.weight_minmax(W, w_bits)
    pw = ops.pack_weight(W, d, z, w_bits)
    out = torch.zeros(M, N, dtype=torch.float16, device=dev)
    gate = torch.ones(1, N, dtype=torch.float32, device=dev)
    kw = dict(epilogue=epi)
    if epi in (ops.EPI_GATE_RESID, ops.EPI_RESID):
        kw.update(resid=out)
    if epi == ops.EPI_GATE_RESID:
        kw.update(gate=gate, rows_per_gate=M)
    gop = 2.0 * M * N * K
    line = "W%d %-16s M %5d N %4d K %4d:" % (w_bits, name, M, N, K)
    for v in (11, 16):
        t = timed(lambda: ops.gemm_i8(qa, pw, out=out, variant=v, **kw))
        line += "  v%d %6.1f us %.2f POPS" % (v, t * 1e6, gop / t / 1e15)
    print(line, flush=True)
