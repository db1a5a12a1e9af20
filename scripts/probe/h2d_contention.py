"""What a host-to-device copy beside the scan kernel costs it: the configs[1] step in a loop while a copy stream
moves `mb` MB per step (h2d: from pinned memory; d2d: device to device; none):
    python scripts/probe/h2d_contention.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detex_amd import synth  # noqa: E402
from detex_amd.hip import HipContext  # noqa: E402

Nc, L, D, k, chunks = 3, 3000, 25, 8, 24
n = Nc * L
M = 372000 * Nc
U, off = synth.detector_bank(1, D, k, n)
x, plants = synth.continuous_day(1001, chunks, M, Nc, U, off, nplant=50)
xt = torch.from_numpy(x).cuda()
ctx = HipContext(0)
ctx.upload_basis(U, off, Nc)
P = ctx.num_windows(M)
ds = torch.empty((chunks, D, (P + 63) // 64 * 64), dtype=torch.float32, device='cuda')[:, :, :P]
thr = torch.full((D,), 0.3, device='cuda')
hist = torch.zeros((D, 400), dtype=torch.int32, device='cuda')
copy = torch.cuda.Stream()
host = torch.empty(104 * 250000, dtype=torch.float32).pin_memory()
dsrc = torch.empty(104 * 250000, dtype=torch.float32, device='cuda')
ddst = torch.empty(104 * 250000, dtype=torch.float32, device='cuda')
steps = 60


def run(kind, mb, pieces=1):
    ne = mb * 250000
    pend = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        if kind != 'none':
            with torch.cuda.stream(copy):
                for p in range(pieces):
                    a, b = ne * p // pieces, ne * (p + 1) // pieces
                    (ddst[a:b]).copy_(host[a:b] if kind == 'h2d' else dsrc[a:b], non_blocking=True)
        f = ctx.subspace_scan(xt, out=ds, thr=thr, hist=hist, lazy=True)[1]
        if pend is not None:
            pend()
        pend = f
    pend()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%-5s %4d MB per step in %2d pieces: %.3f ms per step, scan kernel %.3f ms'
          % (kind, mb, pieces, 1e3 * dt / steps, ctx.kernel_ms_mean(32)))


# the DS tensor of every step in a buffer of its own, round robin over `nbuf` (SubSpace.detex has three units in flight)
for nbuf in (1, 3, 6):
    bufs = [torch.empty((chunks, D, (P + 63) // 64 * 64), dtype=torch.float32, device='cuda')[:, :, :P] for _ in range(nbuf)]
    pend = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        f = ctx.subspace_scan(xt, out=bufs[i % nbuf], thr=thr, hist=hist, lazy=True)[1]
        if pend is not None:
            pend()
        pend = f
    pend()
    torch.cuda.synchronize()
    print('%d DS buffers: %.3f ms per step, scan kernel %.3f ms' % (nbuf, 1e3 * (time.perf_counter() - t0) / steps, ctx.kernel_ms_mean(32)))
    del bufs
# fresh DS tensor per step from the caching allocator (what StationScan does)
pend = None
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    f = ctx.subspace_scan(xt, thr=thr, hist=hist, lazy=True)[1]
    if pend is not None:
        pend()
    pend = f
pend()
torch.cuda.synchronize()
print('fresh DS per step: %.3f ms per step, scan kernel %.3f ms' % (1e3 * (time.perf_counter() - t0) / steps, ctx.kernel_ms_mean(32)))
run('none', 0)
for rep in range(1):
    run('none', 0)
    run('h2d', 104)
    run('h2d', 52)
    run('h2d', 104, 13)
    run('d2d', 104)
