"""Micro-benchmarks of the individual HIP kernels at the 16x512x512 STDiT shapes (GPU box only).
Writes gpurun_out/kernels.json.  Not part of the product; a measurement helper."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import viditq_amd  # noqa
from viditq_amd import ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab"))
import lab  # noqa: E402  (tools/lab: retired variants / probes live outside the product library)

dev = torch.device("cuda:0")
PEAK_I8 = 5.03e15


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    res = {}
    M = 16384
    g = torch.Generator().manual_seed(0)
    for (N, K) in [(1152, 1152), (3456, 1152), (4608, 1152), (1152, 4608)]:
        x = (torch.randn(1, M, K, generator=g)).half().to(dev)
        W = (torch.randn(N, K, generator=g) * 0.03).half().to(dev)
        qa = ops.rowquant(x)
        d, z = ops.weight_minmax(W, 8)
        pw = ops.pack_weight(W, d, z, 8)
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        for variant in (0, 3, 4, 5, 6):
            try:
                t = timeit(lambda: lab.gemm_i8(qa, pw, out=out, variant=variant))
            except Exception as e:  # noqa
                res["gemm_%dx%d_v%d" % (N, K, variant)] = str(e)
                continue
            tops = 2.0 * M * N * K / t
            res["gemm_%dx%d_v%d" % (N, K, variant)] = {"us": t * 1e6, "TOPS": tops / 1e12, "frac": tops / PEAK_I8}
        d4, z4 = ops.weight_minmax(W, 4)
        pw4 = ops.pack_weight(W, d4, z4, 4)
        t = timeit(lambda: lab.gemm_i8(qa, pw4, out=out, variant=0))
        res["gemm_w4_%dx%d_v0" % (N, K)] = {"us": t * 1e6, "TOPS": 2.0 * M * N * K / t / 1e12}
        t = timeit(lambda: ops.rowquant(x))
        res["rowquant_K%d" % K] = {"us": t * 1e6, "GBps": (M * K * 3) / t / 1e9}
    # LN + modulate + quant
    x = torch.randn(1, M, 1152, generator=g).half().to(dev)
    sh = torch.randn(1, 1152, generator=g).float().to(dev)
    t = timeit(lambda: ops.ln_modulate_rowquant(x, sh, sh))
    res["ln_mod_quant"] = {"us": t * 1e6, "GBps": (M * 1152 * 3) / t / 1e9}
    # attention
    H, D, T, S = 16, 72, 16, 1024
    qkv = torch.randn(M, 3 * 1152, generator=g).half().to(dev)
    o = torch.empty((M, 1152), dtype=torch.float16, device=dev)
    ld = 3456
    t = timeit(lambda: ops.attn_fwd(qkv, qkv[:, 1152:], qkv[:, 2304:], o, T, S, S, H, D, S * ld, ld, S * ld, ld,
                                    S * 1152, 1152))
    fl = 4.0 * T * H * S * S * D
    res["attn_spatial"] = {"us": t * 1e6, "TFLOPS": fl / t / 1e12}
    t = timeit(lambda: ops.attn_temporal(qkv, qkv[:, 1152:], qkv[:, 2304:], o, 1, T, S, H, D, ld, 1152))
    res["attn_temporal"] = {"us": t * 1e6, "GBps": (M * 1152 * 2 * 4) / t / 1e9}
    q = torch.randn(M, 1152, generator=g).half().to(dev)
    kv = torch.randn(120, 2304, generator=g).half().to(dev)
    off = torch.tensor([0, 120], dtype=torch.int32, device=dev)
    t = timeit(lambda: ops.attn_fwd(q, kv, kv[:, 1152:], o, 1, M, 0, H, D, M * 1152, 1152, 0, 2304, M * 1152, 1152,
                                    kv_off=off))
    res["attn_cross"] = {"us": t * 1e6, "GBps": (M * 1152 * 2 * 2) / t / 1e9}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "kernels.json"), "w") as f:
        json.dump(res, f, indent=1)
    for k, v in res.items():
        print(k, v)


if __name__ == "__main__":
    main()
