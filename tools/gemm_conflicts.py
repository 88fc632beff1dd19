"""Round 6: where do the GEMM's LDS bank conflicts come from?  (GPU box only; run under
`rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv`.)
Same tile count and epilogue, main loops of 9 / 36 stages (K = 1152 / 4608) and a K = 128 launch that is almost all epilogue:
conflict cycles that do not grow with K belong to the epilogue (slab transposition), the rest to the loop (fragment reads
and DMA landing).  `python tools/gemm_conflicts.py summarize <dir>` prints the per-shape table from the counter CSVs."""
import collections
import csv
import glob
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == "summarize":
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_i8_wide" in r["Kernel_Name"]:
                rows[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # dispatches of one kernel come in the order launched below: group by position modulo the number of shapes
    print("| kernel | grid | n | LDS bank conflict cycles | LDS active cycles | conflict / active | LDS instructions |")
    print("|---|---|---|---|---|---|---|")
    for (k, gsz), cs in sorted(rows.items()):
        n = len(cs["SQ_LDS_IDX_ACTIVE"])
        for lo, hi, tag in ((0, n // 3, "K = 128"), (n // 3, 2 * n // 3, "K = 1152"), (2 * n // 3, n, "K = 4608")):
            m = {c: sum(v[lo:hi]) / max(hi - lo, 1) for c, v in cs.items()}
            print("| %s %s | %s | %d | %.0f | %.0f | %.4f | %.0f |" % (k, tag, gsz, hi - lo, m.get("SQ_LDS_BANK_CONFLICT", 0),
                  m.get("SQ_LDS_IDX_ACTIVE", 0), m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1),
                  m.get("SQ_INSTS_LDS", 0)))
    sys.exit(0)

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viditq_amd  # noqa
from viditq_amd import ops

dev = torch.device("cuda:0")
M, N = 16384, 1152
g = torch.Generator().manual_seed(0)
for K in (128, 1152, 4608):                 # launched in this order, `iters` each: the summary splits dispatches in thirds
    x = torch.randn(1, M, K, generator=g).half().to(dev)
    W = (torch.randn(N, K, generator=g) * 0.03).half().to(dev)
    qa = ops.rowquant(x)
    d, z = ops.weight_minmax(W, 8)
    pw = ops.pack_weight(W, d, z, 8)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    for _ in range(4):
        ops.gemm_i8(qa, pw, out=out)
torch.cuda.synchronize()
print("done")
