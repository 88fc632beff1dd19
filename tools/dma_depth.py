"""L2 -> LDS fill rate of one CU vs bytes in flight and row-chunk width (GPU box only; measurement helper)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab"))
import lab  # noqa: E402  (tools/lab: retired variants / probes live outside the product library)

lib = _lib.load()
out = torch.zeros(512, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
src = torch.randint(-128, 127, ((16384 + 288) * 4608,), dtype=torch.int8, device="cuda")
for stride in (1152, 4608):
    for mode, bkb, depth in [(441, 32, 1), (442, 32, 2), (444, 32, 4), (446, 32, 6), (411, 64, 1), (412, 64, 2), (413, 64, 3),
                             (421, 128, 1), (422, 128, 2)]:
        it = (stride // bkb) * 40
        for _ in range(2):
            lab.lib().vq_probe_stage_rate(mode, src.data_ptr(), stride, it, 256, out.data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lab.lib().vq_probe_stage_rate(mode, src.data_ptr(), stride, it, 256, out.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        byts = it * 544 * bkb
        print("stride %4d  chunk %3d B  depth %d: %.3f us per 34816 B  = %.1f GB/s per CU, %.1f TB/s chip" % (
            stride, bkb, depth, ms * 1e3 / it * 64 / bkb, byts / ms / 1e6, byts * 256 / ms / 1e9), flush=True)
