"""Known-byte-count kernels for calibrating FETCH_SIZE / WRITE_SIZE (tools/pmc_traffic.sh)."""
import torch
a = torch.empty(16384, 4608, dtype=torch.float16, device="cuda")   # 151 MB
b = torch.empty_like(a)
for _ in range(3):
    a.fill_(1.0)          # writes 151 MB
    b.copy_(a)            # reads 151 MB, writes 151 MB
torch.cuda.synchronize()
