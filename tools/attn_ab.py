"""Round 6: attention kernels timed from graph replays (GPU box only).  One process = one library (VIDITQ_LIB selects another
build of the same C ABI); run it alternately over the arms from a shell loop.  Prints one line per shape:
spatial 16 x 1024 (STDiT), image 2 x 4096 (PixArt-Sigma, B = 2), cross 16384 x 120, temporal + quantizer 1024 x 16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops, _lib

dev = torch.device("cuda:0")
H, D = 16, 72
g = torch.Generator().manual_seed(0)
WHICH = sys.argv[1:] or ["spatial", "image", "cross", "temporal"]


def timeit(fn, n=8, reps=10):
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st):
            for _ in range(n):
                fn()
    for _ in range(12):
        gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


tag = os.path.basename(os.path.dirname(_lib.LIB_PATH))
out = []
# --dump=<file> / --cmp=<file>: the spatial / image outputs of a fixed input, saved by one library and compared bit for bit by another
DUMP = next((w[7:] for w in WHICH if w.startswith("--dump=")), None)
CMP = next((w[6:] for w in WHICH if w.startswith("--cmp=")), None)
if DUMP or CMP:
    res = {}
    for name, n_seq, L in (("spatial", 16, 1024), ("image", 2, 4096), ("ragged", 3, 1000)):
        M = n_seq * L
        q = (torch.randn(M, 3 * 1152, generator=torch.Generator().manual_seed(7)) * 1.3).half().to(dev)
        o = torch.zeros((M, 1152), dtype=torch.float16, device=dev)
        ops.attn_fwd(q, q[:, 1152:], q[:, 2304:], o, n_seq, L, L, H, D, L * 3456, 3456, L * 3456, 3456, L * 1152, 1152)
        res[name] = o.cpu()
    if DUMP:
        torch.save(res, DUMP)
        print("%-16s dumped %s" % (tag, DUMP))
    else:
        ref = torch.load(CMP)
        print("%-16s vs %s: %s" % (tag, CMP, ", ".join("%s %s (max |d| %.3g)" % (
            k, "bit-identical" if torch.equal(res[k], ref[k]) else "DIFFERS", float((res[k].float() - ref[k].float()).abs().max())) for k in res)))
    sys.exit(0)
for name, n_seq, L in (("spatial", 16, 1024), ("image", 2, 4096)):
    if name not in WHICH:
        continue
    M = n_seq * L
    bufs = [torch.randn(M, 3 * 1152, generator=g).half().to(dev) for _ in range(3)]
    o = torch.empty((M, 1152), dtype=torch.float16, device=dev)
    ld = 3456
    i = [0]

    def f():
        i[0] = (i[0] + 1) % len(bufs)
        q = bufs[i[0]]
        ops.attn_fwd(q, q[:, 1152:], q[:, 2304:], o, n_seq, L, L, H, D, L * ld, ld, L * ld, ld, L * 1152, 1152)
    t = timeit(f)
    fl = 4.0 * n_seq * L * L * H * D
    out.append("%s %dx%d %.1f us (%.0f TF)" % (name, n_seq, L, t, fl / t / 1e6))
if "cross" in WHICH:
    qs = [torch.randn(16384, 1152, generator=g).half().to(dev) for _ in range(3)]
    kv = torch.randn(120, 2304, generator=g).half().to(dev)
    off = torch.tensor([0, 120], dtype=torch.int32, device=dev)
    o = torch.empty_like(qs[0])
    i = [0]

    def f():
        i[0] = (i[0] + 1) % len(qs)
        # Lk = the bound on every sample's kv length, as QuantAttention.cross passes it (with Lk = 0 the dispatcher cannot
        # know the keys fit two tiles and takes the generic kernel: what this tool timed until round 6, call 20)
        ops.attn_fwd(qs[i[0]], kv, kv[:, 1152:], o, 1, 16384, 120, H, D, 16384 * 1152, 1152, 0, 2304, 16384 * 1152, 1152, kv_off=off)
    out.append("cross 16384x120 %.1f us" % timeit(f))
if "temporal" in WHICH:
    qkvs = [torch.randn(16384, 3456, generator=g).half().to(dev) for _ in range(3)]
    i = [0]

    def f():
        i[0] = (i[0] + 1) % len(qkvs)
        q = qkvs[i[0]]
        ops.attn_temporal_rowquant(q, q[:, 1152:], q[:, 2304:], 1, 16, 1024, H, D, 3456)
    t = timeit(f)
    out.append("temporal+quant 1024x16 %.1f us (%.2f TB/s)" % (t, (16384 * 3456 * 2 + 16384 * 1152) / t / 1e6))
print("%-16s %s" % (tag, " | ".join(out)), flush=True)
