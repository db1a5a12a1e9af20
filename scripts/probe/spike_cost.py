"""What the window statistics' direct re-summation costs when it triggers: configs[1] with K spikes of 1e5 / 1e6 sigma
per station-day (a window whose variance is below 1e-7 of its block's largest window power is summed again directly).
python scripts/probe/spike_cost.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from detex_amd.hip import HipContext
from detex_amd import synth
Nc, L, D, k, chunks = 3, 3000, 25, 8, 24
n, M = Nc * L, 372000 * Nc
U, off = synth.detector_bank(1, D, k, n)
x0, _ = synth.continuous_day(1, chunks, M, Nc, U, off, nplant=50)
thr = torch.full((D,), 0.3, dtype=torch.float32, device='cuda')
ctx = HipContext(0)
ctx.upload_basis(torch.from_numpy(U).cuda(), off, Nc)
P = ctx.num_windows(M)
ds = torch.empty((chunks, D, (P + 63) // 64 * 64), dtype=torch.float32, device='cuda')[:, :, :P]
rs = np.random.RandomState(0)
for (amp, K) in ((0, 0), (1e4, 100), (1e5, 10), (1e5, 1000), (1e6, 10), (1e6, 1000), (0, 0)):
    x = x0.copy()
    if K:
        x.reshape(-1)[rs.randint(0, x.size, K)] += amp
    xt = torch.from_numpy(x).cuda()
    for _ in range(5):
        ctx.subspace_scan(xt, out=ds, thr=thr, lazy=True)[1]()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pend = None
    for _ in range(20):
        f = ctx.subspace_scan(xt, out=ds, thr=thr, lazy=True)[1]
        if pend: pend()
        pend = f
    pend(); torch.cuda.synchronize()
    print('%4d spikes of %7.0e sigma per station-day: %.3f ms per step (scan kernel %.3f)' %
          (K, amp, 1e3 * (time.perf_counter() - t0) / 20, ctx.kernel_ms_mean(20)))
