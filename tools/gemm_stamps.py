"""Per-wave cycle stamps of the full-line ring GEMM (variant 116 = variant 11 + stamps): where a tile's time
goes (prologue / main loop / dequant + slab / store issue / store drain).  GPU box only."""
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
g = torch.Generator().manual_seed(0)
VARIANT = int(sys.argv[1]) if len(sys.argv) > 1 else 116     # 116 = full-line ring (11), 117 = ping-pong (13)
for (N, K) in [(1152, 1152), (4608, 1152), (1152, 4608)]:
    x = torch.randn(1, M, K, generator=g).half().to(dev)
    W = (torch.randn(N, K, generator=g) * 0.03).half().to(dev)
    qa = ops.rowquant(x)
    d, z = ops.weight_minmax(W, 8)
    pw = ops.pack_weight(W, d, z, 8)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    tiles = (M // 256) * (N // 288)
    stamps = torch.zeros(tiles * 8 * 10, dtype=torch.int64, device=dev)
    for _ in range(3):
        lab.gemm_i8(qa, pw, out=out, variant=VARIANT, gate=stamps.view(torch.float32))
    torch.cuda.synchronize()
    s = stamps.view(tiles, 8, 10).cpu().double()
    t0 = s[:, :, 0].min()                       # first wave start on the chip
    names = ["start", "tile0 landed", "main loop end", "params staged", "slab written", "stores issued", "stores done"]
    print("N%d K%d  (cycles of the shader clock, relative to the first wave's start; mean over waves [min..max])" % (N, K))
    for i, nm in enumerate(names):
        v = s[:, :, i] - t0
        print("  %-14s mean %8.0f   [%8.0f .. %8.0f]" % (nm, v.mean(), v.min(), v.max()))
    dur = (s[:, :, 1:7] - s[:, :, 0:6]).mean(dim=(0, 1))
    print("  phase means:", ", ".join("%s %.0f" % (n, float(v)) for n, v in zip(
        ["prologue", "main loop", "barrier+params", "dequant+slab", "store issue", "store drain"], dur)))
    # chip view on the 100 MHz wall clock (synchronised across the chip; 10 ns ticks)
    ws, we = s[:, :, 7].min(dim=1).values, s[:, :, 8].max(dim=1).values
    ratio = float(((s[:, :, 6] - s[:, :, 0]) / (s[:, :, 8] - s[:, :, 7]).clamp(min=1)).median())
    t_first = ws.min()
    o = torch.argsort(ws)
    ws_s, we_s = ws[o], we[o]
    print("  shader cycles per 10 ns tick: %.1f (%.2f GHz); kernel span first start -> last end: %.2f us" % (
        ratio, ratio / 10, float(we.max() - t_first) / 100))
    print("  workgroup starts (us after the first): 1st %.2f, 64th %.2f, 128th %.2f, 256th %.2f; lifetime mean %.2f us" % (
        0.0, float(ws_s[min(63, tiles - 1)] - t_first) / 100, float(ws_s[min(127, tiles - 1)] - t_first) / 100,
        float(ws_s[min(255, tiles - 1)] - t_first) / 100, float((we - ws).mean()) / 100))
    if tiles > 256:
        for r in range(1, tiles // 256):
            print("  round %d: starts %.2f .. %.2f us, previous round ends %.2f .. %.2f us" % (
                r, float(ws_s[256 * r] - t_first) / 100, float(ws_s[256 * r + 255] - t_first) / 100,
                float(torch.sort(we)[0][256 * (r - 1)] - t_first) / 100, float(torch.sort(we)[0][256 * r - 1] - t_first) / 100))
