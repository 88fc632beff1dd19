"""Launch each GEMM variant a few times at one shape (for rocprofv3 --pmc / --kernel-trace runs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab"))
import lab  # noqa: E402  (tools/lab: retired variants / probes live outside the product library)

dev = torch.device("cuda:0")
M = 16384
N, K = int(sys.argv[1]), int(sys.argv[2])
variants = [int(v) for v in sys.argv[3].split(",")]
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
g = torch.Generator().manual_seed(0)
x = torch.randn(1, M, K, generator=g).half().to(dev)
W = (torch.randn(N, K, generator=g) * 0.03).half().to(dev)
qa = ops.rowquant(x)
d, z = ops.weight_minmax(W, 8)
pw = ops.pack_weight(W, d, z, 8)
out = torch.empty((M, N), dtype=torch.float16, device=dev)
for v in variants:
    for _ in range(iters):
        lab.gemm_i8(qa, pw, out=out, variant=v)
torch.cuda.synchronize()
print("done")
