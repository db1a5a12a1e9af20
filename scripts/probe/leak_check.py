"""Device / host memory over repeated SubSpace.detex calls (8 stations x 3 days, day units, SQLite on):
    python scripts/probe/leak_check.py [calls]"""
import os
import resource
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detex_amd import ContinuousFetcher, EventFetcher, createCluster, createSubSpace, synth  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 15
SR, NC, L, NFAM, NPER, NSTA, NDAY = 100.0, 3, 3300, 25, 8, 8, 3
rs = np.random.RandomState(0)
td = synth.event_families(1, NFAM, NPER, NC, L, maxshift=20, snr=8.0)
stas = ['TA.S%02dA' % i for i in range(NSTA)]
ef = EventFetcher(SR, ['HHE', 'HHN', 'HHZ'])
for sta in stas:
    for e in range(len(td)):
        ef.add(sta, 'ev%03d' % e, td[e].reshape(L, NC).T, starttime=1000.0 + 100 * e, origintime=990.0 + 100 * e, magnitude=1.5)
T1 = int(24 * 3600 * SR)
day = rs.standard_normal((NC, T1)).astype(np.float32)
for k in range(50):
    e, t = int(rs.randint(len(td))), int(rs.randint(L, T1 - 2 * L))
    day[:, t:t + L] += 3.0 * td[e].reshape(L, NC).T / np.std(td[e])
rec = torch.from_numpy(np.tile(day, (1, NDAY))).pin_memory()
cf = ContinuousFetcher(SR, ['HHE', 'HHN', 'HHZ'])
for sta in stas:
    cf.add(sta, rec.numpy(), starttime=1499997600.0)
cf.pinned = {sta: rec for sta in stas}
cl = createCluster(CCreq=0.5, fetch_arg=ef, device=0, saveclust=False)
ss = createSubSpace(Pf=1e-10, clust=cl, conDatFetcher=cf)
ss.attachPickTimes(startSeconds=1.0, defaultDuration=30.0)
ss.SVD(selectCriteria=4, selectValue=7, threshold=0.3, useSingles=False)
os.environ['DETEX_AMD_DS_BUDGET_MB'] = '1024'
with tempfile.TemporaryDirectory() as tmp:
    for i in range(calls):
        res = ss.detex(subspaceDB=os.path.join(tmp, 'S.db'), estimateMags=True, device=0)
        torch.cuda.synchronize()
        if i % 3 == 2 or i == 0:
            import threading
            print('call %2d: %d rows, device allocated %.2f GB reserved %.2f GB, host max RSS %.2f GB, threads %d'
                  % (i, len(res['ss_df']), torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9,
                     resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, threading.active_count()))
