"""Round 5: overlap inside a wave (tools/lab/gemm_sp.hip) - a persistent workgroup on 256 x 192 tiles that feeds a finished tile's
LDS transposition and stores into the NEXT tile's main loop and requests the next tile's first stages before its epilogue -
against the product kernel (interior and general form) and the non-persistent 256 x 192 tile (gemm_loader.hip mode 2):
bit-identity, back-to-back times, ablations.  Timing: every candidate of a shape in turn, ROUNDS times over, after a warm-up
long enough for the clocks to settle - the median per candidate (a candidate timed first after the host-side set-up of a
shape reads 10-25 % slow: the first version of this tool, and tools/gemm_loader.py, did that to the product kernel).
GPU box only."""
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
    """cands: [(name, fn)] -> {name: median us per launch}; the candidates alternate inside every round"""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in cands]
    for _ in range(6):                                   # ~50 ms of continuous launches before anything is timed
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


SHAPES = [(3456, 1152, ops.EPI_NONE, "qkv"), (4608, 1152, ops.EPI_NONE, "fc1"), (1152, 1152, ops.EPI_GATE_RESID, "proj+gate"),
          (1152, 4608, ops.EPI_GATE_RESID, "fc2+gate"), (1152, 1152, ops.EPI_RESID, "cross-proj")]
SP_VARIANTS = ((0, "stores behind the DMA issue"), (32, "stores in front of the stage barrier"), (36, "32 without next-tile prefetch"),
               (64, "non-issuing waves store both groups"), (68, "64 without next-tile prefetch"))
SP_ABLATIONS = ((2, "no dequantisation / transposition / stores"), (8, "no global stores"), (16, "no slab writes"), (24, "neither"))
for N, K, epi, name in SHAPES:
    x = (torch.randn(1, M, K, generator=g) * 1.5).half().to(dev)
    W = (torch.randn(N, K, generator=g) * 0.04).half().to(dev)
    b = (torch.randn(N, generator=g) * 0.1).float().to(dev)
    qa = ops.rowquant(x)
    d, z = ops.weight_minmax(W, 8)
    pw = ops.pack_weight(W, d, z, 8)
    res = (torch.randn(M, N, generator=g)).half().to(dev)
    gate = (torch.rand(1, N, generator=g) + 0.5).float().to(dev)
    kw = dict(epilogue=epi, bias=b)
    if epi == ops.EPI_GATE_RESID:
        kw.update(resid=res, gate=gate, rows_per_gate=M)
    if epi == ops.EPI_RESID:
        kw.update(resid=res)
    ref = ops.gemm_i8(qa, pw, variant=11, **kw)
    variants = SP_VARIANTS if epi == ops.EPI_NONE else SP_VARIANTS[:1]
    same = {}
    for v, _ in variants:
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
        lab.gemm_sp(qa, pw, out=out, variant=v, **kw)
        torch.cuda.synchronize()
        same[v] = torch.equal(out, ref)
    cands = [("product, interior form", lambda: ops.gemm_i8(qa, pw, **kw)),
             ("product, general form", lambda: ops.gemm_i8(qa, pw, variant=11, **kw)),
             ("lab copy of the interior form", lambda: lab.gemm_loader(qa, pw, mode=1, **kw)),
             ("256x192 non-persistent", lambda: lab.gemm_loader(qa, pw, mode=2, **kw))]
    for v, nm in variants:
        cands.append(("pipelined: " + nm, (lambda v_: (lambda: lab.gemm_sp(qa, pw, variant=v_, **kw)))(v)))
    if epi == ops.EPI_NONE:
        for v, nm in SP_ABLATIONS:
            cands.append(("pipelined, ablation: " + nm, (lambda v_: (lambda: lab.gemm_sp(qa, pw, variant=v_)))(v)))
    r = bench(cands)
    print("%-10s N %4d K %4d  (median of %d alternating bursts of %d, us)" % (name, N, K, ROUNDS, BURST))
    for nm, _ in cands:
        v = [vv for vv, n2 in variants if "pipelined: " + n2 == nm]
        print("    %-72s %6.1f%s" % (nm, r[nm], ("  bit-identical" if same[v[0]] else "  DIFFERS") if v else ""), flush=True)
