"""Attention yardstick (run under rocprofv3 --kernel-trace --stats): torch.nn.functional.scaled_dot_product_attention
(the ROCm flash / memory-efficient backends) at the block's spatial shape [16 sequences, 16 heads, 1024 tokens, 72 dims]
and at PixArt-Sigma's [1, 16, 4096, 72]."""
import torch
import torch.nn.functional as F
dev = torch.device("cuda:0")
for (n, L) in ((16, 1024), (1, 4096)):
    q, k, v = [torch.randn(n, 16, L, 72, device=dev).half() for _ in range(3)]
    for _ in range(30):
        o = F.scaled_dot_product_attention(q, k, v)
    torch.cuda.synchronize()
