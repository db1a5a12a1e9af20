"""Pinned host -> device copy rate of one station-day (107 MB) over 1 / 2 / 4 copy streams."""
import time

import torch

n = 24 * 1116000
src = torch.empty(n, dtype=torch.float32).pin_memory()
dst = torch.empty(n, dtype=torch.float32, device='cuda')
for ns in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    per = (n + ns - 1) // ns
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(10):
            for k, s in enumerate(streams):
                with torch.cuda.stream(s):
                    dst[k * per:(k + 1) * per].copy_(src[k * per:(k + 1) * per], non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
    print('%d stream(s): %.2f ms per 107 MB = %.1f GB/s' % (ns, dt * 1e3, n * 4 / dt / 1e9))
