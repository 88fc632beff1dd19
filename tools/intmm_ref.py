"""What a tuned library GEMM reaches at the block's shapes (GPU box only): torch._int_mm (hipBLASLt int8 -> int32) and
fp16 matmul, timed back to back; under `rocprofv3 --kernel-trace --stats` the kernel names show the macro-tile /
MFMA shape the library picked.  Yardstick only - nothing here is used by the product."""
import sys, torch
dev = torch.device("cuda:0")
M = 16384


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
    return e0.elapsed_time(e1) / n * 1e-3


for N, K in [(1152, 1152), (3456, 1152), (4608, 1152), (1152, 4608)]:
    a = torch.randint(-128, 127, (M, K), dtype=torch.int8, device=dev)
    b = torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev)
    bt = torch.randint(-128, 127, (N, K), dtype=torch.int8, device=dev)
    out = torch.empty((M, N), dtype=torch.int32, device=dev)
    try:
        t = timeit(lambda: torch._int_mm(a, b, out=out))
        print("int8  N %4d K %4d  [K,N] layout : %7.1f us  %.2f POPS" % (N, K, t * 1e6, 2.0 * M * N * K / t / 1e15))
    except Exception as e:  # noqa
        print("int_mm [K,N] failed:", str(e)[:100])
    try:
        t = timeit(lambda: torch._int_mm(a, bt.t(), out=out))
        print("int8  N %4d K %4d  [N,K]^T layout: %7.1f us  %.2f POPS" % (N, K, t * 1e6, 2.0 * M * N * K / t / 1e15))
    except Exception as e:  # noqa
        print("int_mm [N,K]^T failed:", str(e)[:100])
    ah = torch.randn(M, K, device=dev).half()
    bh = torch.randn(N, K, device=dev).half()
    oh = torch.empty((M, N), dtype=torch.float16, device=dev)
    t = timeit(lambda: torch.matmul(ah, bh.t(), out=oh))
    print("fp16  N %4d K %4d               : %7.1f us  %.2f PFLOP/s" % (N, K, t * 1e6, 2.0 * M * N * K / t / 1e15))
