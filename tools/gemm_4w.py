"""Four waves of 128 x 144 - or, with `--waves 12`, twelve of 64 x 96 - (tools/lab/gemm_4w.hip) against the product ring kernel
(8 waves of 64 x 144): bit-identity, back-to-back times at the block's GEMM shapes, the lab kernel's ablations and its
per-wave cycle stamps.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab"))
import lab  # noqa: E402

NWV = 12 if "--waves" in sys.argv and sys.argv[sys.argv.index("--waves") + 1] == "12" else 4
VARIANTS = (0,) if NWV == 12 else (0, 1)
dev = torch.device("cuda:0")
M = 16384
g = torch.Generator().manual_seed(0)


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
    for v in VARIANTS:
        out = lab.gemm_4w(qa, pw, variant=v, waves=NWV, **kw)
        same = torch.equal(out, ref)
        line += " | %dw v%d %6.1f us %s" % (NWV, v, timeit(lambda: lab.gemm_4w(qa, pw, variant=v, waves=NWV, **kw)), "bit-identical" if same else
                                           "DIFFERS (max %g)" % float((out.float() - ref.float()).abs().max()))
    print(line, flush=True)
    if epi == ops.EPI_NONE:
        print("    ablations of the %d-wave kernel (us): " % NWV + ", ".join(
            "%s %.1f" % (nm, timeit(lambda: lab.gemm_4w(qa, pw, variant=v, waves=NWV)))
            for v, nm in ((101, "no DMA after prologue"), (108, "no fragment reads"), (109, "neither"), (102, "no MFMA"))), flush=True)
        tiles = (M // 256) * (N // 288)
        if NWV == 12:        # the stamped build takes the general epilogue, which half slabs do not have (it traps)
            continue
        for nm, fn, nw in (("ring 8 waves", lambda st: lab.gemm_i8(qa, pw, variant=116, gate=st.view(torch.float32)), 8),
                           ("%d waves" % NWV, lambda st: lab.gemm_4w(qa, pw, variant=116, waves=NWV, gate=st.view(torch.float32)), NWV)):
            stamps = torch.zeros(tiles * nw * 10, dtype=torch.int64, device=dev)
            for _ in range(3):
                fn(stamps)
            torch.cuda.synchronize()
            s = stamps.view(tiles, nw, 10).cpu().double()
            dur = (s[:, :, 1:7] - s[:, :, 0:6]).mean(dim=(0, 1))
            ratio = float(((s[:, :, 6] - s[:, :, 0]) / (s[:, :, 8] - s[:, :, 7]).clamp(min=1)).median())
            print("    stamps %-12s (%.2f GHz): " % (nm, ratio / 10) + ", ".join("%s %.0f" % (n, float(v)) for n, v in zip(
                ["prologue", "main loop", "barrier+params", "dequant+slab", "store issue", "store drain"], dur)), flush=True)
