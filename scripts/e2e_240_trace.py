"""Steady state of SubSpace.detex over day units, for a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/t240 -- python3 scripts/e2e_240_trace.py [days] [raw]
    python scripts/e2e_240_trace.py analyse <kernel_trace.csv> [units]
The analysis takes the last `units` scan-kernel launches of the trace: period per unit, GPU busy time, the
kernels' shares and the largest idle gaps."""
import csv
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == 'analyse':
    rows = list(csv.DictReader(open(sys.argv[2])))
    units = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                 r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:40]) for r in rows)
    scans = [e for e in ev if e[2].startswith(('ds_fft_pp_kernel', 'ds_fft_kernel'))]      # (not the re-scan of listed blocks)
    t0, t1 = scans[-units - 1][0], scans[-1][0]
    win = [e for e in ev if t0 <= e[0] < t1]
    busy, end, gaps = 0, win[0][0], []
    for s, e, n in win:
        if s > end:
            gaps.append((s - end, n))
            end = s
        if e > end:
            busy += e - end
            end = e
    print('%d units: period %.3f ms, GPU busy %.3f ms per unit, %d kernels per unit'
          % (units, (t1 - t0) / 1e6 / units, busy / 1e6 / units, len(win) // units))
    tot = {}
    for s, e, n in win:
        tot[n] = tot.get(n, (0, 0))
        tot[n] = (tot[n][0] + (e - s), tot[n][1] + 1)
    for n, (v, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:16]:
        print('   %-42s %8.1f us per unit  (%5.2f launches)' % (n, v / 1e3 / units, c / units))
    gs = {}
    for g, n in gaps:
        gs[n] = gs.get(n, 0) + g
    print('idle before kernel (us per unit):')
    for n, g in sorted(gs.items(), key=lambda kv: -kv[1])[:8]:
        print('   %-42s %8.1f' % (n, g / 1e3 / units))
    # what runs between the end of one unit's scan kernel and the start of the next one's: mean start (relative to
    # the scan's end), duration and queue of every kernel in that interval, in order of start
    per = {}
    inter = []
    for a, b in zip(scans[-units - 1:-1], scans[-units:]):
        inter.append((b[0] - a[1]) / 1e3)
        for r in rows:
            st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            if a[1] <= st < b[0] or (st < a[1] < en and not r['Kernel_Name'].count('ds_fft')):
                nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:36]
                key = (nm, r.get('Queue_Id', '?'))
                per.setdefault(key, []).append(((st - a[1]) / 1e3, (en - st) / 1e3))
    print('scan end -> next scan start: %.1f us mean (min %.1f max %.1f)' % (sum(inter) / len(inter), min(inter), max(inter)))
    for (nm, q), v in sorted(per.items(), key=lambda kv: sum(x[0] for x in kv[1]) / len(kv[1])):
        print('   %-38s queue %-3s  start %+8.1f us  duration %7.1f us  (%.2f per unit)'
              % (nm, q, sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v), len(v) / units))
    # three consecutive units as they ran
    print('columns:', ', '.join(rows[0].keys()))
    a0, a1 = scans[-units // 2][0], scans[-units // 2 + 3][1]
    for r in sorted(rows, key=lambda r: int(r['Start_Timestamp'])):
        st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if a0 <= st <= a1:
            print('   %9.1f us  %8.1f us  queue %-2s stream %-3s %s' % ((st - a0) / 1e3, (en - st) / 1e3, r.get('Queue_Id', '?'), r.get('Stream_Id', '?'),
                                                        r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:44]))
    sys.exit(0)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from detex_amd import ContinuousFetcher, EventFetcher, createCluster, createSubSpace, synth  # noqa: E402

NDAY = int(sys.argv[1]) if len(sys.argv) > 1 else 4
RAW = len(sys.argv) > 2 and sys.argv[2] == 'raw'
SR, NC, L, NFAM, NPER, NSTA = 100.0, 3, 3300, 25, 8, int(os.environ.get('E2E_STATIONS', '8'))
MAGS = os.environ.get('E2E_MAGS', '1') == '1'
PIN = os.environ.get('E2E_PIN', '1') == '1'
rs = np.random.RandomState(0)
td = synth.event_families(1, NFAM, NPER, NC, L, maxshift=20, snr=8.0)
stas = ['TA.S%02dA' % i for i in range(NSTA)]
ef = EventFetcher(SR, ['HHE', 'HHN', 'HHZ'])
for sta in stas:
    for e in range(len(td)):
        ef.add(sta, 'ev%03d' % e, td[e].reshape(L, NC).T, starttime=1000.0 + 100 * e, origintime=990.0 + 100 * e,
               magnitude=1.0 + (e % 5) * .2)
T1 = int(24 * 3600 * SR)
day = rs.standard_normal((NC, T1)).astype(np.float32)
for k in range(50):
    e, t = int(rs.randint(len(td))), int(rs.randint(L, T1 - 2 * L))
    day[:, t:t + L] += 3.0 * td[e].reshape(L, NC).T / np.std(td[e])
rec = torch.empty((NC, NDAY * T1), dtype=torch.float32).pin_memory()
for d in range(NDAY):
    rec[:, d * T1:(d + 1) * T1] = torch.from_numpy(day)
cf = ContinuousFetcher(SR, ['HHE', 'HHN', 'HHZ'], raw=RAW)
for sta in stas:
    cf.add(sta, rec.numpy(), starttime=1499997600.0)
if PIN:
    cf.pinned = {sta: rec for sta in stas}
_dummies = [torch.cuda.Stream() for _ in range(int(os.environ.get('E2E_DUMMY_STREAMS', '0')))]   # shifts which hardware queue later streams get
cl = createCluster(CCreq=0.5, fetch_arg=ef, device=0, saveclust=False)
ss = createSubSpace(Pf=1e-10, clust=cl, conDatFetcher=cf)
ss.attachPickTimes(startSeconds=1.0, defaultDuration=30.0)
ss.SVD(selectCriteria=4, selectValue=7, threshold=0.3, useSingles=False)
os.environ.setdefault('DETEX_AMD_DS_BUDGET_MB', '1024')
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = ss.detex(subspaceDB=None, estimateMags=MAGS, device=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms = [c.kernel_ms_mean(min(NDAY, 16)) for c in ss._hip_ctx.values()]
    print('call %d: %.1f ms, %.3f ms per station-day, %d detections; scan kernel %.3f ms (mean of the contexts\' last launches)'
          % (rep, 1e3 * dt, 1e3 * dt / (NSTA * NDAY), len(res['ss_df']), sum(kms) / len(kms)))
if os.environ.get('E2E_CPROFILE') == '1':
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    res = ss.detex(subspaceDB=None, estimateMags=True, device=0)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
