"""Cross attention A/B (GPU box only): VQ_ATTN_CROSS=reg selects the register-resident kernel, default the LDS-resident one;
each arm is its own process (the switch is read once).  16384 queries x 16 heads x 72 against 120 / 80 / 40 / 20 prompt tokens,
rotating q buffers (HBM-cold), plus the max deviation from an fp32 softmax reference at 120 keys."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops
dev = torch.device("cuda:0")
H, D = 16, 72
g = torch.Generator().manual_seed(0)
qs = [torch.randn(16384, 1152, generator=g).half().to(dev) for _ in range(6)]
kv = torch.randn(120, 2304, generator=g).half().to(dev)
o = torch.empty_like(qs[0])
i = [0]


def timeit(fn, n=200, warm=30):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run(Lk):
    i[0] = (i[0] + 1) % len(qs)
    off = offs[Lk]
    ops.attn_fwd(qs[i[0]], kv[:Lk], kv[:Lk, 1152:], o, 1, 16384, Lk, H, D, 16384 * 1152, 1152, 0, 2304, 16384 * 1152, 1152, kv_off=off)


offs = {Lk: torch.tensor([0, Lk], dtype=torch.int32, device=dev) for Lk in (120, 80, 40, 20)}
line = "VQ_ATTN_CROSS=%s:" % os.environ.get("VQ_ATTN_CROSS", "lds")
for Lk in (120, 80, 40, 20):
    line += "  Lk %d %.1f us" % (Lk, timeit(lambda: run(Lk)))
q = qs[0]
ops.attn_fwd(q, kv, kv[:, 1152:], o, 1, 16384, 120, H, D, 16384 * 1152, 1152, 0, 2304, 16384 * 1152, 1152, kv_off=offs[120])
qf = q.float().reshape(16384, H, D).permute(1, 0, 2)
kf = kv[:, :1152].float().reshape(120, H, D).permute(1, 0, 2)
vf = kv[:, 1152:].float().reshape(120, H, D).permute(1, 0, 2)
ref = (torch.softmax(qf @ kf.transpose(1, 2) / D ** 0.5, dim=-1) @ vf).permute(1, 0, 2).reshape(16384, 1152)
line += "  rel-L2 vs fp32 softmax %.2e" % float((o.float() - ref).norm() / ref.norm())
print(line)
