import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd
from viditq_amd import _lib
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab"))
import lab  # noqa: E402  (tools/lab: retired variants / probes live outside the product library)
lib = _lib.load()
out = torch.zeros(512, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
iters, blocks = 2000, 256
for mode, name in [(0, "36x mfma16x16x64"), (1, "+13 ds_read_b128"), (2, "+barrier"), (3, "+reads+barrier"),
                   (4, "18x mfma32x32x32"), (5, "32x32 +reads"), (7, "32x32 +reads+barrier"),
                   (9, "reads | sched_barrier | mfma"), (11, "same + barrier"), (22, "asm: reads+mfma all VGPR"), (20, "asm: frags in AGPR"), (21, "asm: acc in AGPR")]:
    for _ in range(2):
        lab.lib().vq_probe_mfma_rate(mode, iters, blocks, out.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lab.lib().vq_probe_mfma_rate(mode, iters, blocks, out.data_ptr(), st); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    ops = blocks * 8 * iters * 36 * 2 * 16 * 16 * 64
    print("%-26s %.3f ms  %.0f TOPS  (%.2f us / iteration)" % (name, ms, ops / ms / 1e9, ms * 1e3 / iters))

src = torch.randint(-128, 127, ((16384 + 288) * 4608,), dtype=torch.int8, device="cuda")
names = {100: "LDS-DMA only", 110: "LDS-DMA + reads", 101: "LDS-DMA + MFMA", 111: "LDS-DMA + reads + MFMA",
         200: "global_load(regs) only", 201: "global_load + MFMA", 211: "global_load + reads + MFMA",
         301: "MFMA only", 311: "reads + MFMA"}
for stride in (1152,):
    for mode in (301, 311, 100, 110, 101, 111, 200, 201, 211):
        it = 72 * 20
        for _ in range(2):
            lab.lib().vq_probe_stage_rate(mode, src.data_ptr(), stride, it, 256, out.data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lab.lib().vq_probe_stage_rate(mode, src.data_ptr(), stride, it, 256, out.data_ptr(), st); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print("stride %4d  %-28s %.3f us / k-tile" % (stride, names[mode], ms * 1e3 / it))
