"""Does a second context on a stream of its own hide one scan's preparation and post pass under the other's
scan kernel?  The configs[1] step alternating between NCTX contexts (same basis, own workspaces):
    python scripts/probe/two_ctx.py [nctx] [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detex_amd import synth  # noqa: E402
from detex_amd.hip import HipContext  # noqa: E402

nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
Nc, L, D, k, chunks = 3, 3000, 25, 8, 24
n = Nc * L
M = 372000 * Nc
U, off = synth.detector_bank(1, D, k, n)
x, plants = synth.continuous_day(1001, chunks, M, Nc, U, off, nplant=50)
xt = torch.from_numpy(x).cuda()
ctxs = []
for i in range(nctx):
    with torch.cuda.stream(torch.cuda.Stream()):
        c = HipContext(0)
    c.upload_basis(U, off, Nc)
    ctxs.append(c)
P = ctxs[0].num_windows(M)
ldp = (P + 63) // 64 * 64
dss = [torch.empty((chunks, D, ldp), dtype=torch.float32, device='cuda')[:, :, :P] for _ in range(nctx)]
thr = torch.full((D,), 0.3, device='cuda')
hists = [torch.zeros((D, 400), dtype=torch.int32, device='cuda') for _ in range(nctx)]
torch.cuda.synchronize()


def run(steps):
    pend = []
    for i in range(steps):
        j = i % nctx
        f = ctxs[j].subspace_scan(xt, out=dss[j], thr=thr, hist=hists[j], lazy=True)[1]
        pend.append(f)
        if len(pend) > nctx:
            pend.pop(0)()
    res = None
    for f in pend:
        res = f()
    return res


t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    run(4)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    res = run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cb, cd, cp, cv = res['cand']
    found = sum(bool(((cb == b) & (cd == d) & (np.abs(cp - p) <= 2)).any()) for (b, p, d) in plants)
    print('nctx %d: %.3f ms per step (%.4g windows/s), kernel %.3f ms, planted %d/%d'
          % (nctx, 1e3 * dt / steps, chunks * P * steps / dt, ctxs[0].kernel_ms_mean(min(16, steps // nctx)), found, len(plants)))
