"""The thirteen GEMM launches of one STDiT block-sample (16384 tokens), each timed back to back (200 launches after
a 50-launch warm-up), for the product kernel; `--pitch4` forces the general (bounds-checked) epilogue through a row
pitch of N + 4 for an A/B of the interior fast path in one binary.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops

dev = torch.device("cuda")
M = 16384
g = torch.Generator().manual_seed(0)
pad = 4 if "--pitch4" in sys.argv else 0
variant = -1
for a_ in sys.argv[1:]:
    if a_.startswith("--variant="):
        variant = int(a_.split("=")[1])
w_bits = 4 if "--w4" in sys.argv else 8
SHAPES = [(3456, 1152, ops.EPI_NONE, "qkv", 2), (1152, 1152, ops.EPI_NONE, "cross-q", 1), (1152, 1152, ops.EPI_GATE_RESID, "proj+gate", 2),
          (1152, 1152, ops.EPI_RESID, "cross-proj", 1), (4608, 1152, ops.EPI_GELU, "fc1+gelu", 1), (1152, 4608, ops.EPI_GATE_RESID, "fc2+gate", 1)]
tot_t, tot_op = 0.0, 0.0
for N, K, epi, name, count in SHAPES:
    x = (torch.randn(1, M, K, generator=g) * 1.5).half().to(dev)
    W = (torch.randn(N, K, generator=g) * 0.04).half().to(dev)
    qa = ops.rowquant(x)
    d, z = ops.weight_minmax(W, w_bits)
    pw = ops.pack_weight(W, d, z, w_bits)
    out = torch.zeros(M, N + pad, dtype=torch.float16, device=dev)
    gate = torch.ones(1, N, dtype=torch.float32, device=dev)
    kw = dict(epilogue=epi, variant=variant)
    if epi in (ops.EPI_GATE_RESID, ops.EPI_RESID):
        kw.update(resid=out)
    if epi == ops.EPI_GATE_RESID:
        kw.update(gate=gate, rows_per_gate=M)
    for _ in range(50):
        ops.gemm_i8(qa, pw, out=out, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(200):
        ops.gemm_i8(qa, pw, out=out, **kw)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 200 * 1e-3
    gop = 2.0 * M * N * K
    tot_t += t * count
    tot_op += gop * count
    print("%-12s N %4d K %4d: %6.1f us  %.2f POPS (%.1f %% of 5.03)" % (name, N, K, t * 1e6, gop / t / 1e15, gop / t / 5.03e13))
print("block-sample GEMM total (13 launches + kv): %.1f us  %.2f POPS = %.1f %%" % (tot_t * 1e6, tot_op / tot_t / 1e15, tot_op / tot_t / 5.03e13))
