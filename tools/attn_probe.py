import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd
from viditq_amd import ops
dev = torch.device("cuda:0")
M, H, D, T, S = 16384, 16, 72, 16, 1024
g = torch.Generator().manual_seed(0)
qkv = torch.randn(M, 3 * 1152, generator=g).half().to(dev)
o = torch.empty((M, 1152), dtype=torch.float16, device=dev)
ld = 3456
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    ops.attn_fwd(qkv, qkv[:, 1152:], qkv[:, 2304:], o, T, S, S, H, D, S * ld, ld, S * ld, ld, S * 1152, 1152)
torch.cuda.synchronize()
print("done")
