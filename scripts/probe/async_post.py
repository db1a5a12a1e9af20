"""Would the post pass of step i on a stream of its own, beside the preparation of step i + 1, shorten the step?
Emulated with two contexts: A runs dtx_subspace_ds (preparation + scan kernel), B runs dtx_ds_post of the same DS
tensor on another stream, ordered by events; two DS buffers in turn.   python scripts/probe/async_post.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detex_amd import synth  # noqa: E402
from detex_amd.hip import HipContext  # noqa: E402

Nc, L, D, k, chunks = 3, 3000, 25, 8, 24
n = Nc * L
M = 372000 * Nc
steps = 60
U, off = synth.detector_bank(1, D, k, n)
x, plants = synth.continuous_day(1001, chunks, M, Nc, U, off, nplant=50)
xt = torch.from_numpy(x).cuda()
A = HipContext(0)
A.upload_basis(U, off, Nc)
sB = torch.cuda.Stream()
with torch.cuda.stream(sB):
    B = HipContext(0)
P = A.num_windows(M)
dss = [torch.empty((chunks, D, (P + 63) // 64 * 64), dtype=torch.float32, device='cuda')[:, :, :P] for _ in range(2)]
thr = torch.full((D,), 0.3, device='cuda')
hist = torch.zeros((D, 400), dtype=torch.int32, device='cuda')
main = torch.cuda.current_stream()


def serial(steps):
    pend = None
    for i in range(steps):
        f = A.subspace_scan(xt, out=dss[i % 2], thr=thr, hist=hist, lazy=True)[1]
        if pend is not None:
            pend()
        pend = f
    pend()


def split(steps):
    post_done = [None, None]
    for i in range(steps):
        j = i % 2
        if post_done[j] is not None:
            main.wait_event(post_done[j])           # the scan may overwrite the buffer once its post pass is through
        A.subspace_ds(xt, out=dss[j])
        ev = torch.cuda.Event()
        ev.record(main)
        sB.wait_event(ev)
        with torch.cuda.stream(sB):
            B.lib.dtx_ds_post  # noqa: B018
            r = B.ds_post.__wrapped__(B, dss[j], thr=None, hist=hist, want_max=True) if False else None
            hist_, maxds, thr_t, key, val, cnt = B._post_buffers(chunks, D, dss[j].device, thr, hist, True, 1 << 20)
            import ctypes as C
            rc = B.lib.dtx_ds_post(B.h, C.c_void_p(dss[j].data_ptr()), P, int(dss[j].stride(1)), chunks, D,
                                   C.c_void_p(hist_.data_ptr()), C.c_void_p(maxds.data_ptr()), C.c_void_p(thr_t.data_ptr()),
                                   C.c_void_p(key.data_ptr()), C.c_void_p(val.data_ptr()), 1 << 20, C.c_void_p(cnt.data_ptr()))
            assert rc == 0
            e2 = torch.cuda.Event()
            e2.record(sB)
        post_done[j] = e2


for name, fn in (('serial (dtx_subspace_scan)', serial), ('post pass on its own stream', split), ('serial', serial),
                 ('post pass on its own stream', split)):
    fn(6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    torch.cuda.synchronize()
    print('%-32s %.3f ms per step, scan kernel %.3f ms' % (name, 1e3 * (time.perf_counter() - t0) / steps, A.kernel_ms_mean(32)))
