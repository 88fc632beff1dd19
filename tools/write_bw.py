import torch
x = torch.empty(16384, 1152, dtype=torch.float16, device="cuda")
y = torch.empty(16384, 3456, dtype=torch.float16, device="cuda")
z = torch.empty_like(y)
def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
a = t(lambda: x.fill_(1.0)); print("fill 37.7MB: %.2f us  %.2f TB/s" % (a, x.numel()*2/a/1e6))
a = t(lambda: y.fill_(1.0)); print("fill 113MB: %.2f us  %.2f TB/s" % (a, y.numel()*2/a/1e6))
a = t(lambda: z.copy_(y)); print("copy 113MB: %.2f us  %.2f TB/s (r+w)" % (a, 2*y.numel()*2/a/1e6))
