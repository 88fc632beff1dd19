"""Per-wave phase stamps of the spatial attention kernel (VQ_ATTN_ABL=128: stamps go into the output buffer)."""
import os, sys, torch
PP = "--pp" in sys.argv
os.environ["VQ_ATTN_STAMP" if PP else "VQ_ATTN_ABL"] = "1" if PP else "128"
if not PP:
    os.environ["VQ_ATTN_PP"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops
dev = torch.device("cuda:0")
M, H, D, T, S = 16384, 16, 72, 16, 1024
g = torch.Generator().manual_seed(0)
qkv = torch.randn(M, 3 * 1152, generator=g).half().to(dev)
EXTRA = (256 * 8 * 16 * 8 + 1024 * 8 * 8) * 8 // (1152 * 2) + 8
o = torch.zeros((M + EXTRA, 1152), dtype=torch.float16, device=dev)
ld = 3456
for _ in range(3):
    o.zero_()
    ops.attn_fwd(qkv, qkv[:, 1152:], qkv[:, 2304:], o, T, S, S, H, D, S * ld, ld, S * ld, ld, S * 1152, 1152)
torch.cuda.synchronize()
sb = o[M:].reshape(-1).view(torch.int64)
st = sb[:256 * 8 * 16 * 8].reshape(256, 8, 16, 8).cpu().double()
names = ["even half-step work", "even barrier wait", "odd half-step work", "odd barrier wait", "-"] if PP else ["QK issue (+loads)", "store_tile", "max/rescale", "exp + PV issue", "barrier wait"]
d = st[:, :, :, 1:6] - st[:, :, :, 0:5]
tile = st[:, :, 1:, 0] - st[:, :, :-1, 0]
print("tile period (start to start), cycles: mean %.0f  min %.0f  max %.0f" % (tile.mean(), tile.min(), tile.max()))
for grp_ in ((0, 4), (4, 8)) if PP else ((0, 8),):
  print("waves", grp_)
  for i, n in enumerate(names[:4] if PP else names):
    v = d[:, grp_[0]:grp_[1], 1:15, i]
    print("  %-18s mean %6.0f   p10 %6.0f  p90 %6.0f" % (n, v.mean(), v.flatten().kthvalue(int(v.numel() * 0.1)).values, v.flatten().kthvalue(int(v.numel() * 0.9)).values))
if PP:
    # group A runs smpv in the odd half-step: slots 2 (start) -> 5 (max / rescale done) -> 6 (first 32 keys: exp + P.V issued) -> 3 (end)
    A = st[:, 0:4, 1:15, :]
    print("group A softmax + P.V: max/rescale %.0f, exp + PV sub-tile 0 %.0f, sub-tile 1 %.0f" % (
        (A[..., 5] - A[..., 2]).mean(), (A[..., 6] - A[..., 5]).mean(), (A[..., 3] - A[..., 6]).mean()))
    sys.exit(0)
w0 = d[0, :, 3, :]
print("workgroup 0, tile 3, per wave:\n", w0)

nwg = 1024
w = sb[256 * 8 * 16 * 8: 256 * 8 * 16 * 8 + nwg * 8 * 8].reshape(nwg, 8, 8).cpu().double()
print("per workgroup (cycles): prologue mean %.0f [p90 %.0f], tile loop mean %.0f, output store %.0f" % (
    (w[:, :, 1] - w[:, :, 0]).mean(), (w[:, :, 1] - w[:, :, 0]).flatten().kthvalue(int(nwg * 8 * 0.9)).values, (w[:, :, 2] - w[:, :, 1]).mean(),
    (w[:, :, 3] - w[:, :, 2]).mean()))
t0 = w[:, :, 6].min()
st, en = (w[:, :, 6].min(dim=1).values - t0) / 100, (w[:, :, 7].max(dim=1).values - t0) / 100
o_ = torch.argsort(st)
print("workgroup start us: 1st %.2f 256th %.2f 257th %.2f 512th %.2f 768th %.2f last %.2f; lifetime mean %.2f us; last end %.2f us" % (
    st[o_[0]], st[o_[255]], st[o_[256]], st[o_[511]], st[o_[767]], st[o_[-1]], (en - st).mean(), en.max()))
