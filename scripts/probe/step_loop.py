"""A/B of the bench loop forms on one box: eager results per step (rounds 1-2), lazy fetch per step,
lazy fetch pipelined one step deep (round 3)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detex_amd import synth  # noqa: E402
from detex_amd.hip import HipContext  # noqa: E402

ctx = HipContext(0)
Nc, L, D, k, chunks = 3, 3000, 25, 8, 24
n, M = Nc * L, 372000 * Nc
U, off = synth.detector_bank(1, D, k, n)
ctx.upload_basis(U, off, Nc)
x, plants = synth.continuous_day(1, chunks, M, Nc, U, off, nplant=50)
xt = torch.from_numpy(x).cuda()
P = ctx.num_windows(M)
ds = torch.empty((chunks, D, (P + 63) // 64 * 64), dtype=torch.float32, device='cuda')[:, :, :P]
thr = np.full(D, 0.3, np.float32)
thr_dev = torch.from_numpy(thr).cuda()
hist = torch.zeros((D, 400), dtype=torch.int32, device='cuda')
steps = 20


def eager(t):
    km = []
    for _ in range(5):
        ctx.subspace_scan(xt, out=ds, thr=t, hist=hist)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.subspace_scan(xt, out=ds, thr=t, hist=hist)
        km.append(ctx.last_kernel_ms())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, np.mean(km)


def lazy(pipelined):
    km = []
    for _ in range(5):
        ctx.subspace_scan(xt, out=ds, thr=thr_dev, hist=hist, lazy=True)[1]()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pend = None
    for _ in range(steps):
        f = ctx.subspace_scan(xt, out=ds, thr=thr_dev, hist=hist, lazy=True)[1]
        if pipelined:
            if pend is not None:
                pend()
            km.append(ctx.last_kernel_ms())
            pend = f
        else:
            f()
            km.append(ctx.last_kernel_ms())
    if pend is not None:
        pend()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, np.mean(km)


for rep in range(3):
    print('eager, host thresholds      %.3f ms/step  kernel %.3f' % eager(thr))
    print('eager, device thresholds    %.3f ms/step  kernel %.3f' % eager(thr_dev))
    print('lazy fetch per step         %.3f ms/step  kernel %.3f' % lazy(False))
    print('lazy fetch, one step deep   %.3f ms/step  kernel %.3f' % lazy(True))
