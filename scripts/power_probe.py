#!/usr/bin/env python
"""Does the scan kernel's time depend on the DATA?  (GPU box)
Times the bench-shaped scan (25 x 8 basis vectors, 24 chunks) on streams that toggle the
matrix pipe differently: N(0,1) noise, all zeros, a constant, and noise with the low 16
mantissa bits cleared (lo parts of the split are all zero).  The work is identical in all
four; only the operand bit patterns change -- a spread in time is the power limit."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detex_amd.hip import HipContext
from detex_amd import synth

Nc, L, D, k, chunks = 3, 3000, 25, 8, 24
n, M = Nc * L, 372000 * Nc
ctx = HipContext(0)
U, det_off = synth.detector_bank(1, D, k, n)
ctx.upload_basis(torch.from_numpy(U).cuda(), det_off, Nc)
P = ctx.num_windows(M)
ds = torch.empty((chunks, D, (P + 63) // 64 * 64), dtype=torch.float32, device='cuda')[:, :, :P]
rng = np.random.RandomState(0)
noise = rng.standard_normal((chunks, M)).astype(np.float32)
trunc = (noise.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
streams = [('noise N(0,1)', noise), ('zeros', np.zeros_like(noise)),
           ('constant 1.0', np.ones_like(noise)), ('noise, bf16-exact (lo parts zero)', trunc)]
for eng in ('bf16x3', 'fp32'):
    ctx.set_engine(eng)
    for name, x in streams:
        xt = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        ms = []
        for it in range(4):
            ctx.subspace_ds(xt, out=ds)
            torch.cuda.synchronize()
            ms.append(ctx.last_kernel_ms())
        print('%-7s %-36s kernel ms %s' % (eng, name, ' '.join('%.2f' % m for m in ms)), flush=True)
