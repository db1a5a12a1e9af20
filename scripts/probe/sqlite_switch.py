"""e2e_detex_240's SQLite leg under different GIL switch intervals (the writer thread and the main thread trade the
GIL: the default 5 ms interval can hold the scan's enqueueing thread off for that long)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from detex_amd import ContinuousFetcher, EventFetcher, createCluster, createSubSpace, synth
SR, NC, L, NFAM, NPER, NSTA, NDAY = 100.0, 3, 3300, 25, 8, 8, 30
rs = np.random.RandomState(0)
td = synth.event_families(1, NFAM, NPER, NC, L, maxshift=20, snr=8.0)
stas = ['TA.S%02dA' % i for i in range(NSTA)]
ef = EventFetcher(SR, ['HHE', 'HHN', 'HHZ'])
for sta in stas:
    for e in range(len(td)):
        ef.add(sta, 'ev%03d' % e, td[e].reshape(L, NC).T, starttime=1000.0 + 100 * e, origintime=990.0 + 100 * e, magnitude=1.0)
T1 = int(24 * 3600 * SR)
day = rs.standard_normal((NC, T1)).astype(np.float32)
for k in range(50):
    e = int(rs.randint(len(td))); t = int(rs.randint(L, T1 - 2 * L))
    day[:, t:t + L] += 3.0 * td[e].reshape(L, NC).T / np.std(td[e])
rec = torch.empty((NC, NDAY * T1), dtype=torch.float32).pin_memory()
for d in range(NDAY):
    rec[:, d * T1:(d + 1) * T1] = torch.from_numpy(day)
cf = ContinuousFetcher(SR, ['HHE', 'HHN', 'HHZ'])
for sta in stas:
    cf.add(sta, rec.numpy(), starttime=1499997600.0)
cf.pinned = {sta: rec for sta in stas}
cl = createCluster(CCreq=0.5, fetch_arg=ef, device=0, saveclust=False)
ss = createSubSpace(Pf=1e-10, clust=cl, conDatFetcher=cf)
ss.attachPickTimes(startSeconds=1.0, defaultDuration=30.0)
ss.SVD(selectCriteria=4, selectValue=7, threshold=0.3, useSingles=False)
with tempfile.TemporaryDirectory() as tmp:
    for db in (None, os.path.join(tmp, 'S.db')):
        for iv in (0.005, 0.0005, 0.00005):
            sys.setswitchinterval(iv)
            for _ in range(2):
                ss.detex(subspaceDB=db, estimateMags=True, device=0)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); ss.detex(subspaceDB=db, estimateMags=True, device=0); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            print('db=%s switchinterval %g: %s ms per station-day' % (bool(db), iv, ['%.3f' % (1e3 * t / 240) for t in ts]))
