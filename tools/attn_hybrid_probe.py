"""What the costed-only hybrid of DESIGN 5d would gain and cost, MEASURED on the shipping spatial-attention kernel with two
profiling switches (a -DVQ_LAB_ABLATIONS build; results wrong by design): VQ_ATTN32_ABL=8 drops the 64 matrix-pipe cycles per
64-key tile that 16 x 16 x 32 MFMAs over dims 64..79 would save, =16 adds the 16 cross-lane moves per tile that a second lane
layout of P^T would need, =24 both.  One process per setting (the switch is read once).  GPU box only."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops
dev = torch.device("cuda:0")
H, D = 16, 72
g = torch.Generator().manual_seed(0)
res = []
for name, n_seq, L in (("spatial 16 x 1024", 16, 1024), ("pixart 1 x 4096", 1, 4096)):
    M = n_seq * L
    qkv = torch.randn(M, 3 * 1152, generator=g).half().to(dev)
    o = torch.empty((M, 1152), dtype=torch.float16, device=dev)
    ld = 3456
    fn = lambda: ops.attn_fwd(qkv, qkv[:, 1152:], qkv[:, 2304:], o, n_seq, L, L, H, D, L * ld, ld, L * ld, ld, L * 1152, 1152)
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(100):
        fn()
    e1.record()
    torch.cuda.synchronize()
    res.append("%s %.1f us" % (name, e0.elapsed_time(e1) * 10))
print("VQ_ATTN32_ABL=%s: %s" % (os.environ.get("VQ_ATTN32_ABL", "0"), ", ".join(res)))
