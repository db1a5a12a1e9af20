"""Ad-hoc wider sweep: reuse the committed sweep tests with many more seeds."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_gpu_random_sweep as t
from detex_amd.hip import HipContext
ctx = HipContext(0)
bad = 0
for seed in range(24, 24 + int(sys.argv[1])):
    try:
        for eng in ('mfma', 'fft'):
            t.test_random_scan(ctx, seed, eng)
    except Exception as e:
        bad += 1
        print('SCAN seed', seed, repr(e)[:300])
        for tag, c in (('same ctx', ctx), ('new ctx', HipContext(0))):
            try:
                t.test_random_scan(c, seed, 'fft')
                print('   rerun', tag, 'ok')
            except Exception as e2:
                print('   rerun', tag, repr(e2)[:300])
for seed in range(8, 8 + int(sys.argv[2])):
    try:
        t.test_random_xcorr(ctx, seed)
    except Exception as e:
        bad += 1
        print('XCORR seed', seed, repr(e)[:300])
print('failures:', bad)

# optional third argument: seeds of long-event cross-correlations (frequency-domain path)
if len(sys.argv) > 3:
    import numpy as np, torch
    from oracle import detex_oracle as orc
    from detex_amd import synth
    badx = 0
    for seed in range(int(sys.argv[3])):
        rs = np.random.RandomState(5000 + seed)
        Nc = int(rs.choice([1, 2, 3, 4]))
        L = int(rs.randint(1024, 9000))
        N = int(rs.randint(3, 12))
        td = synth.event_families(seed, 2, (N + 1) // 2, Nc, L, maxshift=max(1, L // 30))[:N]
        td = td * (10.0 ** rs.uniform(-4, 4, size=N)).astype(np.float32)[:, None]
        ctx.set_engine('auto')
        cc, lag, ss = ctx.xcorr_allpairs(torch.from_numpy(td).cuda(), Nc)
        ctx.sync()
        cc, lag = cc.cpu().numpy(), lag.cpu().numpy()
        n = Nc * L
        trunc = n // (2 * Nc) - 1
        k = 0
        for b in range(N):
            for c in range(b + 1, N):
                raw = orc.ccx2_raw_fft(td[b].astype(np.float64), td[c].astype(np.float64), Nc)
                rc, rl, rss, res = orc.ccx2_reduce(raw, n, Nc, trunc, 1e-5)
                fin = np.sort(res[np.isfinite(res)])
                ok = abs(cc[k] - rc) <= 1e-5
                if ok and len(fin) > 1 and fin[-1] - fin[-2] > 5e-6 and abs(abs(rc) - 1) > 2e-5:
                    ok = lag[k] == rl
                if not ok:
                    badx += 1
                    print('XCORR-LONG seed', seed, (Nc, L, N, b, c), cc[k], rc, lag[k], rl)
                k += 1
    print('long-event failures:', badx)
