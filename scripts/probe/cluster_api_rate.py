"""createCluster through the API at growing catalogue sizes: how much host time surrounds dtx_xcorr_allpairs
(the reference builds the same condensed matrices and linkage, construct.py:63-103, 186-187).
python scripts/probe/cluster_api_rate.py [N ...]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detex_amd import EventFetcher, createCluster, synth  # noqa: E402
from detex_amd.hip import HipContext  # noqa: E402

SR, NC, L = 100.0, 3, int(os.environ.get('CL_L', '3000'))      # (CL_L=13000: configs[2]'s events)
ctx = HipContext(0)
for N in [int(a) for a in sys.argv[1:]] or [200, 1000, 2000]:
    td = synth.event_families(1, N // 20, 20, NC, L, maxshift=20, snr=8.0)
    ef = EventFetcher(SR, ['HHE', 'HHN', 'HHZ'])
    for e in range(len(td)):
        ef.add('TA.M17A', 'ev%04d' % e, td[e].reshape(L, NC).T, starttime=1000.0 + 100 * e,
               origintime=990.0 + 100 * e, magnitude=1.0)
    cl = createCluster(CCreq=0.5, fetch_arg=ef, ctx=ctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cl = createCluster(CCreq=0.5, fetch_arg=ef, ctx=ctx)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tdd = torch.from_numpy(np.ascontiguousarray(td)).cuda()
    ctx.xcorr_allpairs(tdd, NC)
    ctx.sync()
    t1 = time.perf_counter()
    ctx.xcorr_allpairs(tdd, NC)
    ctx.sync()
    print('N = %4d: createCluster %.1f ms, dtx_xcorr_allpairs alone %.1f ms' % (N, 1e3 * dt, 1e3 * (time.perf_counter() - t1)))
    if N == 2000:
        pr = cProfile.Profile()
        pr.enable()
        createCluster(CCreq=0.5, fetch_arg=ef, ctx=ctx)
        pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
