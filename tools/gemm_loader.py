"""Round 5: the LDS-DMA issue taken off the MFMA waves (tools/lab/gemm_loader.hip) against the product ring kernel.
mode 0: 256 x 192 tile, 8 MFMA waves of 64 x 96 + 4 dedicated loader waves (3 waves per SIMD, <= 168 VGPRs);
mode 2: the same tile without loader waves (waves 0-3 issue) - the control; mode 1: the product's 256 x 288 tile with waves
0-3 issuing every piece.  Bit-identity, back-to-back times at the block's GEMM shapes, ablations, per-wave stamps.  GPU box.
(The back-to-back columns are timed in order and the first one follows the shape's host-side set-up: it reads slow by 10-25 % -
tools/gemm_sp.py times the same forms alternately and warm; the stamps are cycle counts and do not care.)"""
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
MODES = [(1, "asym 256x288"), (0, "loaders 256x192"), (2, "control 256x192")]
TILE_N = {0: 192, 1: 288, 2: 192}
NWS = {0: 8, 1: 8, 2: 8}


def timeit(fn, n=100, warm=20):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


SHAPES = [(3456, 1152, ops.EPI_NONE, "qkv"), (1152, 1152, ops.EPI_NONE, "cross-q"), (1152, 1152, ops.EPI_GATE_RESID, "proj+gate"),
          (4608, 1152, ops.EPI_NONE, "fc1"), (1152, 4608, ops.EPI_GATE_RESID, "fc2+gate")]
for N, K, epi, name in SHAPES:
    x = (torch.randn(1, M, K, generator=g) * 1.5).half().to(dev)
    W = (torch.randn(N, K, generator=g) * 0.04).half().to(dev)
    qa = ops.rowquant(x)
    d, z = ops.weight_minmax(W, 8)
    pw = ops.pack_weight(W, d, z, 8)
    res = (torch.randn(M, N, generator=g)).half().to(dev)
    gate = (torch.rand(1, N, generator=g) + 0.5).float().to(dev)
    kw = dict(epilogue=epi)
    if epi == ops.EPI_GATE_RESID:
        kw.update(resid=res, gate=gate, rows_per_gate=M)
    ref = ops.gemm_i8(qa, pw, variant=11, **kw)
    line = "%-10s N %4d K %4d: ring(11) %6.1f us" % (name, N, K, timeit(lambda: ops.gemm_i8(qa, pw, variant=11, **kw)))
    for mode, mname in MODES:
        out = lab.gemm_loader(qa, pw, mode=mode, **kw)
        same = torch.equal(out, ref)
        line += " | %s %6.1f us %s" % (mname, timeit(lambda: lab.gemm_loader(qa, pw, mode=mode, **kw)), "bit-identical" if same else
                                       "DIFFERS (max %g)" % float((out.float() - ref.float()).abs().max()))
    print(line, flush=True)
    if epi == ops.EPI_NONE:
        for mode, mname in MODES:
            print("    ablations %-16s (us): " % mname + ", ".join(
                "%s %.1f" % (nm, timeit(lambda: lab.gemm_loader(qa, pw, mode=mode, variant=v)))
                for v, nm in ((101, "no DMA after prologue"), (108, "no fragment reads"), (109, "neither"), (102, "no MFMA"))), flush=True)
        print("    ablations %-16s (us): " % "ring(11)" + ", ".join(
            "%s %.1f" % (nm, timeit(lambda: lab.gemm_i8(qa, pw, variant=v)))
            for v, nm in ((101, "no DMA after prologue"), (108, "no fragment reads"), (109, "neither"), (102, "no MFMA"))), flush=True)
        cases = [("ring 8 waves", lambda st: lab.gemm_i8(qa, pw, variant=116, gate=st.view(torch.float32)), 8, 288)]
        for mode, mname in MODES:
            cases.append((mname, (lambda m_: (lambda st: lab.gemm_loader(qa, pw, mode=m_, variant=116, gate=st.view(torch.float32))))(mode),
                          NWS[mode], TILE_N[mode]))
        for nm, fn, nw, bn in cases:
            tiles = (M // 256) * (N // bn)
            stamps = torch.zeros(tiles * nw * 10, dtype=torch.int64, device=dev)
            for _ in range(3):
                fn(stamps)
            torch.cuda.synchronize()
            s = stamps.view(tiles, nw, 10).cpu().double()
            dur = (s[:, :, 1:7] - s[:, :, 0:6]).mean(dim=(0, 1))
            ratio = float(((s[:, :, 6] - s[:, :, 0]) / (s[:, :, 8] - s[:, :, 7]).clamp(min=1)).median())
            nk = K // 64
            print("    stamps %-16s (%.2f GHz): " % (nm, ratio / 10) + ", ".join("%s %.0f" % (n, float(v)) for n, v in zip(
                ["prologue", "main loop", "barrier+params", "dequant+slab", "store issue", "store drain"], dur)) +
                  "  -> %.0f cycles per 64-byte k-step (MFMA time %d)" % (float(dur[1]) / nk, 36 * (bn // 8) * 16 // 36 * 1 if False else (bn // 32) * 4 * 2 * 16),
                  flush=True)
