import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd
from viditq_amd import ops
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab"))
import lab  # noqa: E402  (tools/lab: retired variants / probes live outside the product library)
dev = torch.device("cuda:0")
M, N, K = 16384, int(sys.argv[1]), int(sys.argv[2])
v = int(sys.argv[3])
g = torch.Generator().manual_seed(0)
x = torch.randn(1, M, K, generator=g).half().to(dev); W = (torch.randn(N, K, generator=g) * 0.03).half().to(dev)
qa = ops.rowquant(x); d, z = ops.weight_minmax(W, 8); pw = ops.pack_weight(W, d, z, 8)
out = torch.empty((M, N), dtype=torch.float16, device=dev)
def t(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
print("N", N, "K", K, "variant", v, "nkt_env", os.environ.get("VQ_GEMM_NKT"), "us %.2f" % t(lambda: lab.gemm_i8(qa, pw, out=out, variant=v)))
