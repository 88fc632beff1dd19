"""Is the shipping GEMM power-limited?  Runs one GEMM shape back to back for ~2 s and samples rocm-smi
(power, sclk) while the queue is busy; prints rate per launch over time.  GPU box only."""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops

dev = torch.device("cuda")
M = 16384
g = torch.Generator().manual_seed(0)
print(subprocess.run("rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -iE 'power|sclk' | head -6",
                     shell=True, capture_output=True, text=True).stdout)
for (N, K, epi, name) in ((3456, 1152, ops.EPI_NONE, "qkv"), (1152, 4608, ops.EPI_GATE_RESID, "fc2+gate+resid"),
                          (1152, 1152, ops.EPI_GATE_RESID, "proj+gate+resid"), (4608, 1152, ops.EPI_GELU, "fc1+gelu")):
    x = (torch.randn(1, M, K, generator=g) * 1.5).half().to(dev)
    W = (torch.randn(N, K, generator=g) * 0.04).half().to(dev)
    qa = ops.rowquant(x)
    d, z = ops.weight_minmax(W, 8)
    pw = ops.pack_weight(W, d, z, 8)
    out = torch.zeros(M, N, dtype=torch.float16, device=dev)
    gate = torch.ones(1, N, dtype=torch.float32, device=dev)
    kw = dict(epilogue=epi)
    if epi == ops.EPI_GATE_RESID:
        kw.update(resid=out, gate=gate, rows_per_gate=M)
    n = 20000 if N * K < 2e6 else 8000
    ev0, ev1, ev2, ev3 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for _ in range(20):
        ops.gemm_i8(qa, pw, out=out, **kw)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(200):
        ops.gemm_i8(qa, pw, out=out, **kw)
    ev1.record()
    for _ in range(n):
        ops.gemm_i8(qa, pw, out=out, **kw)
    ev2.record()
    for _ in range(200):
        ops.gemm_i8(qa, pw, out=out, **kw)
    ev3.record()
    smi = subprocess.run("sleep 0.4; rocm-smi --showpower --showclocks 2>/dev/null | grep -iE 'power|sclk' | head -4",
                         shell=True, capture_output=True, text=True).stdout
    torch.cuda.synchronize()
    gop = 2.0 * M * N * K
    t_first, t_last = ev0.elapsed_time(ev1) / 200, ev2.elapsed_time(ev3) / 200
    print("%-16s first 200: %.1f us (%.2f POPS)   after %.1f s busy: %.1f us (%.2f POPS)" %
          (name, t_first * 1e3, gop / t_first / 1e12, ev0.elapsed_time(ev2) / 1e3, t_last * 1e3, gop / t_last / 1e12))
    print(smi, flush=True)
