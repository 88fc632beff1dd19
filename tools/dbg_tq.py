import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import viditq_amd
from viditq_amd import ops
def h16(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half()
dev = torch.device("cuda:0")
for (T, S, H, D) in [(16, 64, 16, 72), (16, 37, 16, 72), (16, 128, 16, 72)]:
    Cc = H * D
    qkv = h16(T * S, 3 * Cc, seed=T * 31 + S).to(dev)
    o = torch.empty((T * S, Cc), dtype=torch.float16, device=dev)
    ops.attn_temporal(qkv, qkv[:, Cc:], qkv[:, 2 * Cc:], o, 1, T, S, H, D, 3 * Cc, Cc)
    ref = ops.rowquant(o.view(1, T * S, Cc))
    for rep in range(2):
        o2 = torch.zeros_like(o)
        got = ops.attn_temporal_rowquant(qkv, qkv[:, Cc:], qkv[:, 2 * Cc:], 1, T, S, H, D, 3 * Cc, o=o2)
        torch.cuda.synchronize()
        bad_r = (got.R != ref.R).nonzero().flatten()
        bad_x = (got.xq != ref.xq).any(dim=1).nonzero().flatten()
        do = (o2 != o).nonzero()
        print("   fp16 outputs differing:", do.shape[0], do[:4].tolist(), [(float(o[i, j]), float(o2[i, j])) for i, j in do[:4].tolist()])
        print(T, S, H, D, "rep", rep, "sx eq", torch.equal(got.sx, ref.sx), "zx eq", torch.equal(got.zx, ref.zx),
              "bad R rows", bad_r.numel(), bad_r[:8].tolist(), "bad xq rows", bad_x.numel(), bad_x[:8].tolist())
        if bad_r.numel():
            r = int(bad_r[0]); print("   row", r, "t", r // S, "s", r % S, "R got/ref", int(got.R[r]), int(ref.R[r]),
                                     "sum codes got", int(got.xq[r].int().sum()), "ref", int(ref.xq[r].int().sum()))
        if bad_x.numel():
            r = int(bad_x[0]); cols = (got.xq[r] != ref.xq[r]).nonzero().flatten()
            print("   xq row", r, "cols", cols[:12].tolist(), got.xq[r][cols[:6]].tolist(), ref.xq[r][cols[:6]].tolist())
            xrow = o[r].float().cpu(); d = ref.sx[r].cpu(); zp = (ref.zx[r] + 128).float().cpu()
            qo = torch.clamp(torch.round(xrow / d) + zp, 0, 255) - 128
            c0 = int(cols[0]); xv = xrow[c0]
            print("   oracle code", int(qo[c0]), "x", float(xv), "delta", float(d), "x/delta", float(xv / d), "x*inv", float(xv * (1.0 / d)),
                  "oracle==ref row", bool(torch.equal(qo.to(torch.int8), ref.xq[r, :Cc].cpu())), "oracle==got row", bool(torch.equal(qo.to(torch.int8), got.xq[r, :Cc].cpu())))
