"""Round 6: cycle stamps of the phased spatial-attention kernel (attn_fwd64p_kernel, lab build with -DVQ_ATTN_STAMPS=<NB>).
GPU box only.  VIDITQ_LIB must point at the stamped build.  Prints, per wave half (0: leads, 1: one slot behind), the median
length of each phase of tile 1 and of the barrier waits between them, plus the whole loop per slot."""
import ctypes
import os
import sys

import numpy as np
import torch

lib = ctypes.CDLL(os.environ["VIDITQ_LIB"])
dev = torch.device("cuda:0")
H, D = 16, 72
n_seq, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 1024)
M = n_seq * L
q = (torch.randn(M, 3 * 1152, generator=torch.Generator().manual_seed(7)) * 1.3).half().to(dev)
o = torch.zeros((M, 1152), dtype=torch.float16, device=dev)
nwg = 8 * ((n_seq * H + 7) // 8) * ((L + 511) // 512)
st = torch.zeros((nwg, 8, 16), dtype=torch.int32, device=dev)
f = lib.vq_lab_attn64p_stamped
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_long] * 6 + [ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
k, v = q[:, 1152:], q[:, 2304:]
for _ in range(5):
    rc = f(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), n_seq, L, L, H, L * 3456, 3456, L * 3456, 3456, L * 1152, 1152,
           D ** -0.5, st.data_ptr(), None)
    assert rc == 0, rc
torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.int64) & 0xffffffff


def d(a, b):
    return (s[:, :, a] - s[:, :, b]) & 0xffffffff


names = [("prologue", 1, 0), ("loop", 2, 1), ("epilogue", 3, 2), ("V(2)", 5, 4), ("barrier", 6, 5), ("M(3) MFMAs", 7, 6), ("tile wait", 8, 7),
         ("barrier", 9, 8), ("V(3)", 10, 9), ("barrier", 11, 10), ("M(4)", 12, 11), ("barrier", 13, 12), ("one tile (4 slots)", 13, 4)]
nslots = 4 * ((L + 63) // 64) + 2
for half in (0, 1):
    print("waves %d-%d:" % (4 * half, 4 * half + 3))
    for nm, a, b in names:
        x = d(a, b)[:, 4 * half:4 * half + 4].reshape(-1)
        extra = "  (%.0f per slot over %d slots)" % (np.median(x) / nslots, nslots) if nm == "loop" else ""
        print("  %-20s median %7.0f  p10 %7.0f  p90 %7.0f cycles%s" % (nm, np.median(x), np.percentile(x, 10), np.percentile(x, 90), extra))
