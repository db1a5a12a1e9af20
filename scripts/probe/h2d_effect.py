"""What a concurrent host-to-device copy costs the scan (bench.py's pcie_inclusive runs 3.55 ms per step against
3.21 with the day resident, its scan kernel 2.91 against 2.75): the same 40 steps at configs[1] with
  none      no copy at all
  free      a 107 MB pinned copy per step into a buffer nobody reads (no dependency: pure contention)
  dep       the pcie_inclusive pattern (double buffer, the scan waits for its day)
  free/16   the free copy cut into 16 pieces
  d2d       a 107 MB device-to-device copy per step on the side stream instead (the blit kernel's share)
usage: [HSA_ENABLE_SDMA=0] python scripts/probe/h2d_effect.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from detex_amd.hip import HipContext
from detex_amd import synth
Nc, L, D, k, chunks = 3, 3000, 25, 8, 24
n, M = Nc * L, 372000 * Nc
U, off = synth.detector_bank(1, D, k, n)
x, _ = synth.continuous_day(1, chunks, M, Nc, U, off, nplant=50)
thr = torch.full((D,), 0.3, dtype=torch.float32, device='cuda')
ctx = HipContext(0)
ctx.upload_basis(torch.from_numpy(U).cuda(), off, Nc)
P = ctx.num_windows(M)
ds = torch.empty((chunks, D, (P + 63) // 64 * 64), dtype=torch.float32, device='cuda')[:, :, :P]
xp = torch.from_numpy(x).pin_memory()
bufs = [torch.from_numpy(x).cuda() for _ in range(2)]
spare = torch.empty_like(bufs[0])
copy = torch.cuda.Stream()
main = torch.cuda.current_stream()
ready = [torch.cuda.Event() for _ in range(2)]
done = [torch.cuda.Event() for _ in range(2)]


def run(mode, steps=40):
    pieces = 16 if mode == 'free/16' else 1
    flat_s, flat_p = spare.view(-1), xp.view(-1)
    cut = [(i * flat_p.numel() // pieces, (i + 1) * flat_p.numel() // pieces) for i in range(pieces)]

    def upload(i):
        with torch.cuda.stream(copy):
            if mode == 'dep':
                copy.wait_event(done[i])
                bufs[i].copy_(xp, non_blocking=True)
                ready[i].record(copy)
            elif mode in ('free', 'free/16'):
                for a, b in cut:
                    flat_s[a:b].copy_(flat_p[a:b], non_blocking=True)
            elif mode == 'd2d':
                spare.copy_(bufs[1 - i], non_blocking=True)
    for i in range(2):
        done[i].record(main)
    torch.cuda.synchronize()
    pend = None
    t0 = None
    for i in range(-4, steps):
        if i == 0:
            if pend: pend(); pend = None
            torch.cuda.synchronize(); t0 = time.perf_counter()
        cur = i & 1
        upload(cur if mode == 'dep' else cur)
        if mode == 'dep':
            main.wait_event(ready[cur])
        _, f = ctx.subspace_scan(bufs[cur], out=ds, thr=thr, lazy=True)
        done[cur].record(main)
        if pend: pend()
        pend = f
    pend()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ks = np.array(ctx.kernel_ms_list(min(steps, 64)))
    return 1e3 * dt / steps, float(np.median(ks)), float(ks.min()), float(ks.max())


print('HSA_ENABLE_SDMA=%s' % os.environ.get('HSA_ENABLE_SDMA', '(unset)'))
# bare copy rate
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    spare.copy_(xp, non_blocking=True)
torch.cuda.synchronize()
print('bare H2D: %.2f ms per %.0f MB = %.1f GB/s' % (1e3 * (time.perf_counter() - t0) / 5, x.nbytes / 1e6,
                                                   5 * x.nbytes / (time.perf_counter() - t0) / 1e9))
for mode in ('none', 'free', 'dep', 'free/16', 'd2d', 'none', 'free', 'dep'):
    run(mode, 8)
    print('%-8s ms/step %.3f   scan kernel median %.3f  min %.3f  max %.3f' % ((mode,) + run(mode)))
