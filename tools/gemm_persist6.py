"""Round 6: the persistent form of the interior GEMM (tools/lab/gemm_persist6.hip = gemm_wide_r6.h INT 2 - one workgroup per CU
walks the tiles, half epilogue slabs, the next tile's stage 0 requested before the current tile is dequantised and stored)
against the product's interior form (variant 19) and general form (11) on the block's multi-round launches: bit-identity and back-to-back times
(every candidate of a shape in turn, ROUNDS times over after a warm-up, medians - tools/gemm_sp.py's method).  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab"))
import lab  # noqa: E402

dev = torch.device("cuda:0")
M = 16384
g = torch.Generator().manual_seed(0)
ROUNDS, BURST = 7, 40


def bench(cands):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in cands]
    for _ in range(6):
        for _, fn in cands:
            for _ in range(10):
                fn()
    torch.cuda.synchronize()
    t = {nm: [] for nm, _ in cands}
    for _ in range(ROUNDS):
        for (nm, fn), (e0, e1) in zip(cands, ev):
            for _ in range(5):
                fn()
            e0.record()
            for _ in range(BURST):
                fn()
            e1.record()
        torch.cuda.synchronize()
        for (nm, _), (e0, e1) in zip(cands, ev):
            t[nm].append(e0.elapsed_time(e1) / BURST * 1e3)
    return {nm: sorted(v)[len(v) // 2] for nm, v in t.items()}


SHAPES = [(3456, 1152, ops.EPI_NONE, 8, "qkv"), (4608, 1152, ops.EPI_NONE, 8, "fc1"), (4608, 1152, ops.EPI_GELU, 8, "fc1+gelu"),
          (1152, 4608, ops.EPI_GATE_RESID, 8, "fc2+gate (1 round)"), (1152, 1152, ops.EPI_GATE_RESID, 8, "proj+gate (1 round)"),
          (3456, 1152, ops.EPI_NONE, 4, "qkv W4"), (4608, 1152, ops.EPI_NONE, 4, "fc1 W4")]
for N, K, epi, wb, name in SHAPES:
    x = (torch.randn(1, M, K, generator=g) * 1.5).half().to(dev)
    W = (torch.randn(N, K, generator=g) * 0.04).half().to(dev)
    b = (torch.randn(N, generator=g) * 0.1).float().to(dev)
    qa = ops.rowquant(x)
    d, z = ops.weight_minmax(W, wb)
    pw = ops.pack_weight(W, d, z, wb)
    res = (torch.randn(M, N, generator=g)).half().to(dev)
    gate = (torch.rand(1, N, generator=g) + 0.5).float().to(dev)
    kw = dict(epilogue=epi, bias=b)
    if epi == ops.EPI_GATE_RESID:
        kw.update(resid=res, gate=gate, rows_per_gate=M)
    ref = ops.gemm_i8(qa, pw, variant=11, **kw)
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
    for _ in range(3):
        lab.gemm_persist6(qa, pw, out=out, **kw)
    torch.cuda.synchronize()
    same = torch.equal(out, ref)
    o2 = torch.empty_like(out)
    cands = [("interior (19)", lambda: ops.gemm_i8(qa, pw, variant=19, out=o2, **kw)),
             ("persistent (20)", lambda: lab.gemm_persist6(qa, pw, out=o2, **kw)),
             ("general (11)", lambda: ops.gemm_i8(qa, pw, variant=11, out=o2, **kw))]
    r = bench(cands)
    print("%-20s N %4d K %4d W%d: interior %6.1f us | persistent %6.1f us (%+.1f %%, %s) | general %6.1f us" % (
        name, N, K, wb, r["interior (19)"], r["persistent (20)"], (r["persistent (20)"] / r["interior (19)"] - 1) * 100,
        "bit-identical" if same else "DIFFERS", r["general (11)"]), flush=True)
