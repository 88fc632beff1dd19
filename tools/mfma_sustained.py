"""Is the int8 MFMA rate power-limited when sustained?  Pure-MFMA probe (no memory traffic) launched back to back
for ~3 s; rate per launch over time, plus rocm-smi power / clock readings taken while the queue is busy.  GPU box only."""
import os
import subprocess
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
iters, blocks, n = 20000, 256, 300
ops = blocks * 8 * iters * 36 * 2 * 16 * 16 * 64
for mode, name in ((0, "36 x mfma_i32_16x16x64_i8 per iteration, 8 waves/CU"), (3, "+ 13 ds_read_b128 + barrier")):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        lab.lib().vq_probe_mfma_rate(mode, iters, blocks, out.data_ptr(), st)
        ev[i + 1].record()
    smi = subprocess.run("sleep 1.5; rocm-smi --showpower --showclocks 2>/dev/null | grep -iE 'power|sclk|mclk|fclk' | head -8",
                         shell=True, capture_output=True, text=True).stdout
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    print(name)
    for i in (0, 5, 20, 50, 100, 200, 299):
        print("  launch %3d (t = %6.0f ms): %.2f ms  %.0f TOPS" % (i, sum(ms[:i]), ms[i], ops / ms[i] / 1e9))
    print(smi, flush=True)
