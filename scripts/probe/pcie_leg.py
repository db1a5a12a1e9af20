"""bench.py's pcie_inclusive leg alone, several times in one process (20 / 40 / 80 steps): python scripts/probe/pcie_leg.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from detex_amd.hip import HipContext
from detex_amd import synth
Nc, L, D, k, chunks = 3, 3000, 25, 8, 24
n, M = Nc * L, 372000 * Nc
U, off = synth.detector_bank(1, D, k, n)
x, _ = synth.continuous_day(1, chunks, M, Nc, U, off, nplant=50)
ctx = HipContext(0)
ctx.upload_basis(torch.from_numpy(U).cuda(), off, Nc)
P = ctx.num_windows(M)
ds = torch.empty((chunks, D, (P + 63) // 64 * 64), dtype=torch.float32, device='cuda')[:, :, :P]
hist = torch.zeros((D, 400), dtype=torch.int32, device='cuda')
thr = np.full(D, 0.3, dtype=np.float32)
xt = torch.from_numpy(x).cuda()
thr_dev = torch.from_numpy(thr).cuda()
for _ in range(150):
    ctx.subspace_scan(xt, out=ds, thr=thr_dev, hist=hist, lazy=True)[1]()
for steps in (20, 40, 80, 20, 40, 80):
    r = bench.pcie_inclusive(ctx, x, ds, thr, hist, chunks * P, steps)
    print('steps %3d: %.3f ms per step, scan kernel mean %.3f min %.3f median %.3f max %.3f' %
          (steps, r['ms_per_step'], r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_median'], r['kernel_ms_max']))
