"""Does a pinned H2D copy on a side stream run under a long kernel on the default stream?"""
import time
import torch
x = torch.empty(27 * 1000 * 1000, dtype=torch.float32).pin_memory()        # 108 MB
d = torch.empty_like(x, device='cuda')
a = torch.randn(8192, 8192, device='cuda')
side = torch.cuda.Stream()
torch.cuda.synchronize()
def busy(n):
    for _ in range(n):
        torch.mm(a, a)
for trial in range(3):
    torch.cuda.synchronize()
    t0 = time.time()
    e0, e1, c0, c1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    e0.record()
    busy(20)
    e1.record()
    with torch.cuda.stream(side):
        c0.record(side)
        d.copy_(x, non_blocking=True)
        c1.record(side)
    torch.cuda.synchronize()
    print('kernels %.2f ms; copy start +%.2f ms, copy %.2f ms (%.1f GB/s); wall %.2f ms'
          % (e0.elapsed_time(e1), e0.elapsed_time(c0), c0.elapsed_time(c1), x.numel() * 4 / c0.elapsed_time(c1) / 1e6,
             1e3 * (time.time() - t0)))
# copy alone
torch.cuda.synchronize()
c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
c0.record(); d.copy_(x, non_blocking=True); c1.record(); torch.cuda.synchronize()
print('copy alone %.2f ms' % c0.elapsed_time(c1))
