#!/usr/bin/env python
"""Secondary benchmark (BASELINE.json configs[2]): all-pairs normalised
cross-correlation similarity matrix on one MI355X.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detex_amd.hip import HipContext  # noqa: E402
from detex_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--events', type=int, default=2000)
ap.add_argument('--L', type=int, default=13000)
ap.add_argument('--nc', type=int, default=3)
ap.add_argument('--steps', type=int, default=1)
ap.add_argument('--cpu-pairs', type=int, default=0)
ap.add_argument('--engine', default='auto', choices=['auto', 'fft', 'bf16x3', 'fp32'])
a = ap.parse_args()
N, L, Nc = a.events, a.L, a.nc
n = Nc * L
nfam = max(1, N // 20)
td = synth.event_families(2, nfam, (N + nfam - 1) // nfam, Nc, L)[:N]
ctx = HipContext(0)
ctx.set_engine(a.engine)
t = torch.from_numpy(td).cuda()
cc, lag, ss = ctx.xcorr_allpairs(t, Nc)      # warm-up (allocates workspaces)
ctx.sync()
t0 = time.perf_counter()
for _ in range(a.steps):
    cc, lag, ss = ctx.xcorr_allpairs(t, Nc)
ctx.sync()
dt = (time.perf_counter() - t0) / a.steps
kms = ctx.last_kernel_ms()
npairs = N * (N - 1) // 2
S = L - L // 2
mac = Nc * ((2 * S + 1) * L - S * (S + 1))
flops = 2.0 * mac * npairs
out = {'metric': 'pairs/sec (all-pairs normalised xcorr)', 'value': npairs / dt, 'unit': 'pairs/s',
       'config': {'workload': 'configs[2]: %d events, Nc=%d, L=%d (n=%d), %d pairs' % (N, Nc, L, n, npairs)},
       'seconds': dt,
       'engine': a.engine,
       'roofline': ({'bound': 'mfma', 'kernel': ('xc_mfma_kernel_s<13,2>' if a.engine != 'fp32' else 'xc_mfma_kernel<13,2>') + ' (+ remainder tiles)',
                     'achieved': flops / (kms * 1e-3) / 1e12, 'peak': 157.3 if a.engine == 'fp32' else 2500.0, 'unit': 'TFLOP/s',
                     'frac': flops / (kms * 1e-3) / 1e12 / (157.3 if a.engine == 'fp32' else 2500.0), 'kernel_ms': kms,
                     'flop_per_pair': 2.0 * mac}
                    if a.engine in ('fp32', 'bf16x3') or L < 1024 else
                    {'kernel': 'xc_fft22_kernel', 'kernel_ms': kms,
                     'contraction_equivalent_tflops': flops / (kms * 1e-3) / 1e12,
                     'note': 'frequency-domain lag search (dtx_xcfft.hip); the direct contraction it replaces is %.3g flop per pair' % (2.0 * mac)}),
       'cc_mean': float(cc.mean()), 'cc_max': float(cc.max())}
if a.cpu_pairs:
    from oracle import detex_oracle as orc
    fds = [orc.get_freq_domain(td[i].astype(np.float64)) for i in range(2)]
    t0 = time.perf_counter()
    for _ in range(a.cpu_pairs):
        orc.ccx2_fft(td[0], td[1], Nc, fds[0], fds[1])
    out['cpu_baseline'] = {'value': a.cpu_pairs / (time.perf_counter() - t0), 'unit': 'pairs/s',
                           'cores': 1, 'kind': 'port',
                           'sample': '%d pairs, float64 FFT path' % a.cpu_pairs}
print(json.dumps(out))
