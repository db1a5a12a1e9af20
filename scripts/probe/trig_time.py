"""Duration of dtx_trigger_stats for one station-day's worth of triggers (configs[1] geometry: n = 9000,
25 detectors x 8 vectors, 8 training events each, ~50 triggers):  python scripts/probe/trig_time.py [ntrig]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detex_amd import synth  # noqa: E402
from detex_amd.hip import HipContext  # noqa: E402

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 51
Nc, L, D, k, chunks = 3, 3000, 25, 8, 4
n = Nc * L
M = 372000 * Nc
U, off = synth.detector_bank(1, D, k, n)
x, plants = synth.continuous_day(7, chunks, M, Nc, U, off, nplant=8)
ctx = HipContext(0)
ctx.upload_basis(U, off, Nc)
xt = torch.from_numpy(x).cuda()
ds = ctx.subspace_ds(xt)
rs = np.random.RandomState(0)
P = ds.shape[2]
trig = np.stack([rs.randint(0, chunks, nt), rs.randint(0, D, nt), rs.randint(60000, P - 1000, nt),
                 np.zeros(nt), np.zeros(nt)], axis=1).astype(np.int32)
trig[:, 3] = trig[:, 2]
trig[:, 4] = trig[:, 2] - 250
ewf = torch.randn(D * 8, n, device='cuda')
eoff = torch.arange(0, D * 8 + 1, 8, dtype=torch.int32, device='cuda')
for full in (True, False):
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        out = ctx.trigger_stats(xt, M, M, ds, trig, 1, 500, ewf if full else None, eoff, 8)
        e1.record()
        torch.cuda.synchronize()
    print('full=%s: %.1f us for %d triggers' % (full, 1e3 * e0.elapsed_time(e1), nt))
