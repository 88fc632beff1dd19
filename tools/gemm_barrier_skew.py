"""Round 5: who waits at the stage barrier of the interior GEMM form?  One stamped launch (vq_gemm_i8_stamped) per shape: per wave
the shader cycles from the start of the main loop (first stage landed) to its ARRIVAL at the stage barrier of k-tile 1, and the
end of the main loop.  Waves 0-3 issue the LDS-DMA pieces, waves 4-7 (their SIMD partners) none.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (N, K) in ((3456, 1152), (1152, 1152), (1152, 4608)):
    x = (torch.randn(1, 16384, K, generator=g) * 1.5).half().to(dev)
    W = (torch.randn(N, K, generator=g) * 0.04).half().to(dev)
    qa = ops.rowquant(x)
    d, z = ops.weight_minmax(W, 8)
    pw = ops.pack_weight(W, d, z, 8)
    for _ in range(10):
        ops.gemm_i8(qa, pw, variant=19)
    _, st = ops.gemm_i8_stamped(qa, pw)
    torch.cuda.synchronize()
    s = st.cpu().double()
    arr = (s[:, :, 9] - s[:, :, 1]).mean(dim=0)           # per wave: loop start -> arrival at the barrier of k-tile 1
    loop = (s[:, :, 2] - s[:, :, 1]).mean(dim=0)
    print("N %d K %d  arrival at the k-tile-1 barrier after loop start (cycles, mean over tiles), waves 0-7: %s" % (
        N, K, " ".join("%.0f" % v for v in arr)))
    print("            issuing waves 0-3 mean %.0f, partners 4-7 mean %.0f  -> the partners wait %.0f cycles per stage barrier; main loop %.0f cycles (%d stages)" % (
        float(arr[:4].mean()), float(arr[4:].mean()), float(arr[:4].mean() - arr[4:].mean()), float(loop.mean()), K // 128))
