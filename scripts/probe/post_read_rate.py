"""How fast can the post pass read DS at all?  dtx_ds_post with only the maximum (no histogram,
no candidates) against the full pass, on a configs[1]-sized DS tensor (886 MB)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detex_amd.hip import HipContext  # noqa: E402

ctx = HipContext(0)
nchunk, D, P = 24, 25, 369001
ldp = (P + 63) // 64 * 64
ds = (torch.rand((nchunk, D, ldp), device='cuda') * 0.009)[:, :, :P]
hist = torch.zeros((D, 400), dtype=torch.int32, device='cuda')
maxds = torch.zeros((nchunk, D), dtype=torch.float32, device='cuda')
thr = torch.full((D,), 0.3, device='cuda')
key = torch.empty((1 << 20,), dtype=torch.int64, device='cuda')
val = torch.empty((1 << 20,), dtype=torch.float32, device='cuda')
cnt = torch.zeros((1,), dtype=torch.int32, device='cuda')
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
for name, args in (('max only', (None, maxds, None, None, None, None)),
                   ('max + histogram', (hist, maxds, None, None, None, None)),
                   ('full', (hist, maxds, thr, key, val, cnt))):
    h, m, t, k, v, c = args
    for rep in range(3):
        ctx.timer_start()
        for _ in range(10):
            rc = ctx.lib.dtx_ds_post(ctx.h, p(ds), P, ldp, nchunk, D, p(h), p(m), p(t), p(k), p(v), 1 << 20, p(c))
            assert rc == 0
        ms = ctx.timer_stop() / 10
    print('%-18s %.1f us  %.2f TB/s' % (name, ms * 1e3, nchunk * D * P * 4 / ms / 1e9))
x = ds.contiguous() if False else None
big = torch.empty(886 * 1024 * 1024 // 4, device='cuda')
for rep in range(3):
    ctx.timer_start()
    for _ in range(10):
        s = big.sum()
    ms = ctx.timer_stop() / 10
print('torch.sum of 886 MiB  %.1f us  %.2f TB/s' % (ms * 1e3, big.numel() * 4 / ms / 1e9))
