"""What zero-filled gaps cost the scan (fillZeros=True data): configs[1] with G gaps of 10 s / 60 s per station-day (templates of 30 s), all channels
zeroed.  A window that holds a few live samples at a gap's edge has next to no energy: its block is listed by the
dynamic-range guard and recomputed by the float32 contraction.  python scripts/probe/gap_cost.py"""
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
for (G, glen) in ((0, 0), (10, 1000), (100, 1000), (1, 6000), (10, 6000), (100, 6000), (0, 0)):
    x = x0.copy().reshape(chunks, -1)
    for _ in range(G):
        b, t = int(rs.randint(chunks)), int(rs.randint(0, 372000 - glen))
        x[b, t * Nc:(t + glen) * Nc] = 0.0
    xt = torch.from_numpy(x.reshape(x0.shape)).cuda()
    for _ in range(5):
        ctx.subspace_scan(xt, out=ds, thr=thr, lazy=True)[1]()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pend = None
    for _ in range(20):
        f = ctx.subspace_scan(xt, out=ds, thr=thr, lazy=True)[1]
        if pend: pend()
        pend = f
    pend(); torch.cuda.synchronize()
    print('%4d gaps of %3d s per station-day: %.3f ms per step (scan kernel %.3f)' %
          (G, glen // 100, 1e3 * (time.perf_counter() - t0) / 20, ctx.kernel_ms_mean(20)))
