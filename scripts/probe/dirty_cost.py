"""What dirty data cost the scan at configs[1]: the clean day, the verdict's `config2_dirty` (10 gaps of 60 s + 10
glitches of 1e5 sigma + the planted events), one gap's edge beside an event, 10 / 1000 glitches of 1e5 / 1e6 sigma.
python scripts/probe/dirty_cost.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from detex_amd.hip import HipContext
from detex_amd import synth
Nc, L, D, k, chunks = 3, 3000, 25, 8, 24
n, M = Nc * L, 372000 * Nc
U, off = synth.detector_bank(1, D, k, n)
x0, plants = synth.continuous_day(1, chunks, M, Nc, U, off, nplant=50)
thr = torch.full((D,), 0.3, dtype=torch.float32, device='cuda')
ctx = HipContext(0)
ctx.upload_basis(torch.from_numpy(U).cuda(), off, Nc)
P = ctx.num_windows(M)
ds = torch.empty((chunks, D, (P + 63) // 64 * 64), dtype=torch.float32, device='cuda')[:, :, :P]
def run(x, label):
    xt = torch.from_numpy(x).cuda()
    for _ in range(5):
        ctx.subspace_scan(xt, out=ds, thr=thr, lazy=True)[1]()
    cnt = ctx.repair_counts()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pend = None
    for _ in range(20):
        f = ctx.subspace_scan(xt, out=ds, thr=thr, lazy=True)[1]
        if pend: pend()
        pend = f
    pend(); torch.cuda.synchronize()
    print('%-46s %.3f ms per step (scan kernel %.3f)  listed %d blocks, %d split, %d tiles to the contraction' %
          (label, 1e3 * (time.perf_counter() - t0) / 20, ctx.kernel_ms_mean(20), cnt['blocks'], cnt['split'], cnt['tiles']), flush=True)
run(x0, 'clean')
x = x0.copy(); synth.make_dirty(x, Nc, 11); run(x, 'config2_dirty (10 gaps 60 s + 10 glitches 1e5)')
x = x0.copy(); b, p, d = plants[0]; x[b, (p + L + 20) * Nc:(p + L + 6020) * Nc] = 0; run(x, 'one gap right behind a planted event')
for (amp, K) in ((1e5, 10), (1e5, 1000), (1e6, 10), (1e6, 1000)):
    x = x0.copy(); synth.make_dirty(x, Nc, 3, ngap=0, nspike=K, spike_amp=amp); run(x, '%d glitches of %.0e sigma' % (K, amp))
x = x0.copy(); synth.make_dirty(x, Nc, 5, ngap=100, nspike=0); run(x, '100 gaps of 60 s')
x = x0.copy(); x[3, 500000:] *= 3000.0; run(x, 'a loud onset (x 3000 from there on) in one chunk')
for lvl in (1e3, 5e3, 3e4):
    x = x0.copy(); x[5, 600000:] += lvl; run(x, 'a level step of %.0e sigma in mid-chunk (ADVICE r5: DC-offset record)' % lvl)
run(x0, 'clean')
