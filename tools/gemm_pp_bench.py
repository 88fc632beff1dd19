"""Round-3 experiment: the tile-level ping-pong GEMM (tools/lab/gemm_pp.h; lab variants 230 = units drawn from counters,
231 = static unit walk) against the product ring kernel (variant 11) on the launches of one STDiT block-sample (16384
tokens), back to back (100 launches after a 30-launch warm-up), plus its profiling ablations (static walk): 200 = the
kernel itself, 201 no LDS-DMA after the prologue, 208 no fragment reads, 202 no MFMA (204 / 205 / 213 drop the epilogue
micro-ops: the compiler then removes the dead MFMAs too, so they time the DMA stream + barriers only).  GPU box only.
  python tools/gemm_pp_bench.py [--no-abl] [--m=16384]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops

dev = torch.device("cuda")
M = 16384
for a_ in sys.argv[1:]:
    if a_.startswith("--m="):
        M = int(a_.split("=")[1])
w_bits = 4 if "--w4" in sys.argv else 8
g = torch.Generator().manual_seed(0)
SHAPES = [(3456, 1152, ops.EPI_NONE, "qkv", 2), (1152, 1152, ops.EPI_NONE, "cross-q", 1), (1152, 1152, ops.EPI_GATE_RESID, "proj+gate", 2),
          (1152, 1152, ops.EPI_RESID, "cross-proj", 1), (4608, 1152, ops.EPI_GELU, "fc1+gelu", 1), (1152, 4608, ops.EPI_GATE_RESID, "fc2+gate", 1)]


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


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab"))
import lab as lab  # noqa: E402
abl = "--no-abl" not in sys.argv and w_bits == 8
tot = {11: 0.0, 230: 0.0, 231: 0.0}
tot_op = 0.0
for N, K, epi, name, count in SHAPES:
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
    tot_op += gop * count
    line = "%-11s N %4d K %4d:" % (name, N, K)
    for v in (11, 230, 231):
        fn = ops.gemm_i8 if v == 11 else lab.gemm_i8
        t = timed(lambda: fn(qa, pw, out=out, variant=v, **kw))
        tot[v] += t * count
        line += "  v%d %6.1f us %.2f POPS" % (v, t * 1e6, gop / t / 1e15)
    print(line, flush=True)
    if abl and epi in (ops.EPI_NONE, ops.EPI_GATE_RESID):
        line = "            ablations (us):"
        for v in (200, 204, 201, 205, 208, 213, 202):
            t = timed(lambda: lab.gemm_i8(qa, pw, out=out, variant=v, **kw), n=40, warm=10)
            line += "  %d: %6.1f" % (v, t * 1e6)
        print(line, flush=True)
for v in (11, 230, 231):
    print("block-sample GEMM total, variant %d: %.1f us  %.2f POPS = %.1f %% of 5.03" %
          (v, tot[v] * 1e6, tot_op / tot[v] / 1e15, tot_op / tot[v] / 5.03e13))
