"""What the ragged launch of a station-day (1 + 23 + 1 chunks, 1752 live blocks of 1825) costs against the uniform
launch of 24 chunks (1752 blocks): scan-kernel time from the library's event ring."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detex_amd import synth  # noqa: E402
from detex_amd.hip import HipContext  # noqa: E402

Nc, L, D, k, T = 3, 3000, 25, 8, 372000
U, off = synth.detector_bank(1, D, k, Nc * L)
ctx = HipContext(0)
ctx.upload_basis(U, off, Nc)
x = torch.randn(Nc * 8640000 + Nc * T, device='cuda')
thr = torch.full((D,), 0.3, device='cuda')
bounds = [(0, 132000)] + [(120000 + 360000 * i, T) for i in range(23)] + [(8400000, 240000)]
tab = ctx.chunk_table([s * Nc for s, _ in bounds], [c * Nc for _, c in bounds])
for name, kw in (('uniform 24 x 3720 s', dict(M=Nc * T, chunk_stride=Nc * 360000, nchunk=24)),
                 ('ragged 1 + 23 + 1', dict(chunk_tab=tab))):
    ds, f = ctx.subspace_scan(x, thr=thr, lazy=True, **kw)
    for rep in range(3):
        ctx.timer_start()
        for _ in range(20):
            ds, f = ctx.subspace_scan(x, out=ds, thr=thr, lazy=True, **kw)
        ms = ctx.timer_stop() / 20
        print('%-22s %.3f ms per call, scan kernel %.3f ms' % (name, ms, ctx.kernel_ms_mean(20)))
