"""test_both_lag_search_kernels' data (an event riding 1000 above 1e-3 of signal, an event whose first third is a
constant 500): every pair of the device answer beside the float64 oracle's.  python scripts/probe/lowvar_pairs.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import detex_oracle as orc
from detex_amd.hip import HipContext
from detex_amd import synth
ctx = HipContext(0)
for (Nc, L, N) in ((1, 2500, 12), (5, 2000, 9), (9, 1024, 8)):
    td = synth.event_families(L + 1, 2, (N + 1) // 2, Nc, L, maxshift=max(1, L // 40))[:N]
    td[5] = td[2]; td[7] = 0.0
    td[3] = 1000.0 + 1e-3 * td[3]
    td[6, : Nc * (L // 3)] = 500.0
    cc, lag, ss = ctx.xcorr_allpairs(torch.from_numpy(np.ascontiguousarray(td, dtype=np.float32)).cuda(), Nc)
    ctx.sync()
    cc, lag, ss = cc.cpu().numpy(), lag.cpu().numpy(), ss.cpu().numpy()
    n = Nc * L; trunc = n // (2 * Nc) - 1
    print('Nc %d L %d N %d: pairs with a NaN subsample shift on the device: %d' % (Nc, L, N, int(np.isnan(ss).sum())))
    k = 0
    for b in range(N):
        for c in range(b + 1, N):
            if 3 in (b, c) or 6 in (b, c) or np.isnan(ss[k]):
                raw = orc.ccx2_raw_fft(td[b].astype(np.float64), td[c].astype(np.float64), Nc)
                rc, rl, rss, res = orc.ccx2_reduce(raw, n, Nc, trunc, 1e-5)
                flag = '' if (abs(cc[k] - rc) <= 1e-5 and (lag[k] == rl)) else '   <-- differs'
                print('   (%d,%d) device cc %.6f lag %d ss %.4f | oracle cc %.6f lag %d ss %.4f%s' % (b, c, cc[k], lag[k], ss[k], rc, rl, rss, flag))
            k += 1
