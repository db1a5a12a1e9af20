"""Where does the PCIe-inclusive step (bench.py pcie_inclusive: 4.4 ms against 3.4 ms resident) spend its time?"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detex_amd import synth  # noqa: E402
from detex_amd.hip import HipContext  # noqa: E402

ctx = HipContext(0)
Nc, L, D, k, chunks = 3, 3000, 25, 8, 24
n, M = Nc * L, 372000 * Nc
U, off = synth.detector_bank(1, D, k, n)
ctx.upload_basis(U, off, Nc)
x = np.random.RandomState(0).standard_normal((chunks, M)).astype(np.float32)
xp = torch.from_numpy(x).pin_memory()
bufs = [torch.empty(x.shape, dtype=torch.float32, device='cuda') for _ in range(2)]
bufs[0].copy_(xp); bufs[1].copy_(xp)
P = ctx.num_windows(M)
ds = torch.empty((chunks, D, (P + 63) // 64 * 64), dtype=torch.float32, device='cuda')[:, :, :P]
thr = np.full(D, 0.3, np.float32)
hist = torch.zeros((D, 400), dtype=torch.int32, device='cuda')
copy = torch.cuda.Stream()
main = torch.cuda.current_stream()
steps = 30


def run(name, lazy, upload, thr_dev=None):
    t = thr if thr_dev is None else thr_dev
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            cur = i & 1
            if upload:
                ev = torch.cuda.Event()
                with torch.cuda.stream(copy):
                    bufs[1 - cur].copy_(xp, non_blocking=True)
                    ev.record(copy)
            if lazy:
                _, f = ctx.subspace_scan(bufs[cur], out=ds, thr=t, hist=hist, lazy=True)
                f()
            else:
                ctx.subspace_scan(bufs[cur], out=ds, thr=t, hist=hist)
            if upload:
                main.wait_event(ev)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    print('%-44s %.3f ms per step' % (name, dt * 1e3))


run('resident, eager results', False, False)
run('resident, lazy fetch', True, False)
run('resident, lazy fetch, thresholds on device', True, False, torch.from_numpy(thr).cuda())
run('upload beside the scan, eager results', False, True)
run('upload beside the scan, lazy fetch', True, True)
