"""Streaming yardstick for the quantizer kernels (run under rocprofv3 --kernel-trace --stats): a plain elementwise
fp16 -> int8 conversion moves exactly the quantizer's bytes (2 B read + 1 B written per element) with no reduction."""
import torch
dev = torch.device("cuda:0")
for C in (1152, 4608):
    x = torch.randn(16384, C, device=dev).half()
    out = torch.empty((16384, C), dtype=torch.int8, device=dev)
    for _ in range(50):
        out.copy_(x)
    torch.cuda.synchronize()
