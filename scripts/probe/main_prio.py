"""The pipelined bench loop with the context's own stream at high priority (its side, post-pass and trigger streams
stay at the default): do the scan's preparation kernels then win the CUs from the post pass that runs beside them?
    python scripts/probe/main_prio.py [0|-1]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detex_amd import synth  # noqa: E402
from detex_amd.hip import HipContext  # noqa: E402

prio = int(sys.argv[1]) if len(sys.argv) > 1 else 0
st = torch.cuda.Stream(priority=prio)
with torch.cuda.stream(st):
    ctx = HipContext(0)
    Nc, L, D, k, chunks = 3, 3000, 25, 8, 24
    n, M = Nc * L, 372000 * Nc
    U, off = synth.detector_bank(1, D, k, n)
    ctx.upload_basis(U, off, Nc)
    x, plants = synth.continuous_day(1, chunks, M, Nc, U, off, nplant=50)
    xt = torch.from_numpy(x).cuda()
    P = ctx.num_windows(M)
    ds = torch.empty((chunks, D, (P + 63) // 64 * 64), dtype=torch.float32, device='cuda')[:, :, :P]
    thr_dev = torch.from_numpy(np.full(D, 0.3, np.float32)).cuda()
    hist = torch.zeros((D, 400), dtype=torch.int32, device='cuda')
    for rep in range(4):
        steps, km, pend = 40, [], None
        for _ in range(5):
            ctx.subspace_scan(xt, out=ds, thr=thr_dev, hist=hist, lazy=True)[1]()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            f = ctx.subspace_scan(xt, out=ds, thr=thr_dev, hist=hist, lazy=True)[1]
            if pend is not None:
                pend()
            km.append(ctx.last_kernel_ms())
            pend = f
        pend()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        print('priority %d: %.3f ms/step  kernel %.3f  step - kernel %.3f' % (prio, dt, np.mean(km), dt - np.mean(km)))
