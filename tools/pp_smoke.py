"""First contact of the ping-pong GEMM with hardware: smallest problems first (one unit, two units, one per CU), each
launch followed by a synchronise and a comparison with variant 11.  Run under `timeout`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops

dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
for variant in (31, 30):
    for (M, N, K) in [(256, 144, 1152), (256, 288, 1152), (512, 1152, 1152), (256, 144, 4608), (8192, 1152, 1152), (16384, 3456, 1152),
                      (16384, 1152, 4608)]:
        x = (torch.randn(1, M, K, generator=g) * 1.5).half().to(dev)
        W = (torch.randn(N, K, generator=g) * 0.04).half().to(dev)
        qa = ops.rowquant(x)
        d, z = ops.weight_minmax(W, 8)
        pw = ops.pack_weight(W, d, z, 8)
        ref = ops.gemm_i8(qa, pw, variant=11)
        torch.cuda.synchronize()
        print("variant", variant, (M, N, K), "launching", flush=True)
        out = ops.gemm_i8(qa, pw, variant=variant)
        torch.cuda.synchronize()
        bad = int((out != ref).sum().item())
        print("   mismatching elements:", bad, "of", out.numel(), flush=True)
        if bad:
            rows = (out != ref).any(dim=1).nonzero().flatten()
            cols = (out != ref).any(dim=0).nonzero().flatten()
            print("   rows", rows[:8].tolist(), "...", rows[-4:].tolist(), "n", rows.numel(), " cols", cols[:8].tolist(), "...",
                  cols[-4:].tolist(), "n", cols.numel(), flush=True)
print("done")
