"""Eager launches of the W4A8 quantizers for a counter pass (usage on a GPU box:
   cd /tmp && rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <out> -o p -- python $R/tools/rq_pmc.py).
Graph replays do not carry per-dispatch counters; the Python wrapper's host time does not matter to them."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops
dev = torch.device("cuda:0")
M = 16384
g = torch.Generator().manual_seed(0)
big = [torch.randn(1, M, 1152, generator=g).half().to(dev) for _ in range(8)]
sh = (torch.randn(1, 1152, generator=g) * 0.3).float().to(dev)
sc = (torch.randn(1, 1152, generator=g) * 0.3).float().to(dev)
sm = [torch.exp(torch.randn(1152, generator=g) * 0.5).float().to(dev) for _ in range(3)]
for i in range(16):
    x = big[i % len(big)]
    ops.rowquant(x)
    ops.ln_modulate_rowquant(x, sh, sc)
    ops.rowquant(x, s=sm[0])
    ops.ln_modulate_rowquant(x, sh, sc, smooth=sm[:1])
    ops.rowquant_multi(x, sm)
    ops.ln_modulate_rowquant(x, sh, sc, smooth=sm)
torch.cuda.synchronize()
