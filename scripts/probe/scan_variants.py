"""Why does the scan kernel take 2.95 ms inside SubSpace.detex and 2.74 in bench.py?  The configs[1] step with
one e2e ingredient at a time:  python scripts/probe/scan_variants.py"""
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
steps = 40
thr = torch.full((D,), 0.3, device='cuda')


def loop(ctx, label, call):
    pend = None
    for _ in range(5):
        call()[1]()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        f = call()[1]
        if pend is not None:
            pend()
        pend = f
    pend()
    torch.cuda.synchronize()
    print('%-58s %.3f ms per step, scan kernel %.3f ms' % (label, 1e3 * (time.perf_counter() - t0) / steps, ctx.kernel_ms_mean(32)))


U, off = synth.detector_bank(1, D, k, n)
x, plants = synth.continuous_day(1001, chunks, M, Nc, U, off, nplant=50)
xt = torch.from_numpy(x).cuda()
ctx = HipContext(0)
ctx.upload_basis(U, off, Nc)
hist = torch.zeros((D, 400), dtype=torch.int32, device='cuda')
loop(ctx, 'bench.py step (random orthonormal bank, rows of M)', lambda: ctx.subspace_scan(xt, thr=thr, hist=hist, lazy=True))
# overlapping chunks: views of one day buffer, stride 3600 s
stride = 360000 * Nc
flat = torch.randn((chunks - 1) * stride + M, device='cuda')
loop(ctx, 'overlapping chunks (stride 3600 s of 3720)', lambda: ctx.subspace_scan(flat, M=M, chunk_stride=stride, nchunk=chunks, thr=thr, hist=hist, lazy=True))
# noise only, no planted events
xn = torch.randn((chunks, M), device='cuda')
loop(ctx, 'noise only', lambda: ctx.subspace_scan(xn, thr=thr, hist=hist, lazy=True))
# the e2e bank: SVD of event families (smooth wavelets), 25 x 8
td = synth.event_families(1, 25, 8, Nc, 3300, maxshift=20, snr=8.0)
rows = []
for f in range(25):
    A = td[8 * f:8 * f + 8, 300:300 + n].astype(np.float64)
    A = A - A.mean(axis=1, keepdims=True)
    rows.append(np.linalg.svd(A.T, full_matrices=False)[0][:, :8].T)
U2 = np.concatenate(rows).astype(np.float32)
ctx.upload_basis(U2, off, Nc)
loop(ctx, 'SVD bank of smooth event families, bench data', lambda: ctx.subspace_scan(xt, thr=thr, hist=hist, lazy=True))
loop(ctx, 'SVD bank, noise only', lambda: ctx.subspace_scan(xn, thr=thr, hist=hist, lazy=True))
loop(ctx, 'SVD bank, overlapping chunks', lambda: ctx.subspace_scan(flat, M=M, chunk_stride=stride, nchunk=chunks, thr=thr, hist=hist, lazy=True))
