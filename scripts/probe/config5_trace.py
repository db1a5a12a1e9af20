"""configs[4] (500 Hz, 6 channels, K = 64, one hour per call / 24 hours per call): the step's kernels, for
rocprofv3 --kernel-trace --stats (step = 2.2 x kernel)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detex_amd import synth  # noqa: E402
from detex_amd.hip import HipContext  # noqa: E402

hours = int(sys.argv[1]) if len(sys.argv) > 1 else 1
Nc, L, K, T = 6, 1000, 64, 1800000
n = Nc * L
U, off = synth.detector_bank(5, 1, K, n)
x, plants = synth.continuous_day(5, 1, Nc * T, Nc, U, off, nplant=10)
xt = torch.from_numpy(x).cuda().reshape(1, -1)
if hours > 1:
    xt = xt.expand(hours, -1).contiguous()
ctx = HipContext(0)
ctx.upload_basis(U, off, Nc)
thr = torch.full((1,), 0.3, dtype=torch.float32, device='cuda')
ds, r = ctx.subspace_scan(xt, thr=thr)
ctx.sync()
for rep in range(3):
    t1 = time.perf_counter()
    for _ in range(10):
        ds, r = ctx.subspace_scan(xt, out=ds, thr=thr)
    ctx.sync()
    dt = (time.perf_counter() - t1) / 10
    print('hours per call %d: %.3f ms per call = %.3f ms per hour; scan kernel %.3f ms' % (hours, 1e3 * dt, 1e3 * dt / hours,
                                                                                       ctx.last_kernel_ms()))
