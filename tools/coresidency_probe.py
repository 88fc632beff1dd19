"""Can an HBM-bound kernel run BESIDE the ring GEMM on the same CUs?  Two streams: A loops a GEMM (qkv shape), B loops the
C = 1152 per-token quantizer; wall time of both together against each alone.  The 256 x 288 tile (variant 11: 8 waves x 234-244
VGPRs and all 160 KB of LDS - nothing else fits on its CU) vs the 128 x 288 tile (variant 16: <= 142 VGPRs, 106 KB of LDS: room
for four more waves per SIMD).  GPU box only."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops
dev = torch.device("cuda:0")
M, N, K = 16384, 3456, 1152
g = torch.Generator().manual_seed(0)
x = (torch.randn(1, M, K, generator=g) * 1.5).half().to(dev)
W = (torch.randn(N, K, generator=g) * 0.04).half().to(dev)
qa = ops.rowquant(x)
d, z = ops.weight_minmax(W, 8)
pw = ops.pack_weight(W, d, z, 8)
out = torch.empty(M, N, dtype=torch.float16, device=dev)
xs = [torch.randn(1, M, 1152, generator=g).half().to(dev) for _ in range(4)]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
NG, NQ = 100, 600


def run(gemm_variant, do_a, do_b):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sA.wait_stream(torch.cuda.current_stream())
    sB.wait_stream(torch.cuda.current_stream())
    if do_a:
        with torch.cuda.stream(sA):
            for _ in range(NG):
                ops.gemm_i8(qa, pw, out=out, variant=gemm_variant)
    if do_b:
        with torch.cuda.stream(sB):
            for i in range(NQ):
                ops.rowquant(xs[i & 3])
    torch.cuda.current_stream().wait_stream(sA)
    torch.cuda.current_stream().wait_stream(sB)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for v in (11, 16):
    run(v, True, True)
    a, b, ab = run(v, True, False), run(v, False, True), run(v, True, True)
    print("GEMM variant %d: GEMM x %d alone %.2f ms, quantizer x %d alone %.2f ms, together %.2f ms (sum %.2f; overlap saves %.0f %% of the quantizer time)"
          % (v, NG, a, NQ, b, ab, a + b, 100 * (a + b - ab) / b))
