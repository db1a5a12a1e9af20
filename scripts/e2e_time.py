"""Wall-clock of the reference workflow through the Python API at config-2 size:
createCluster -> createSubSpace -> SVD -> detex on one station-day (24 h, 3-ch,
100 Hz) with ~25 subspaces x 8 events.  Shows how much host time surrounds the
kernels bench.py times in isolation.  Usage: python scripts/e2e_time.py [hours]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from detex_amd import ContinuousFetcher, EventFetcher, createCluster, createSubSpace  # noqa: E402
from detex_amd import synth  # noqa: E402
from detex_amd.hip import HipContext  # noqa: E402

hours = float(sys.argv[1]) if len(sys.argv) > 1 else 24.0
SR, NC, L, NFAM, NPER = 100.0, 3, 3300, 25, 8
rs = np.random.RandomState(0)


def tick(label, t0):
    torch.cuda.synchronize()
    t1 = time.time()
    print('%-34s %8.1f ms' % (label, 1e3 * (t1 - t0)))
    return t1


t0 = time.time()
td = synth.event_families(1, NFAM, NPER, NC, L, maxshift=20, snr=8.0)
NSTA = int(os.environ.get('E2E_STATIONS', '1'))      # > 1: the steady-state cost per station-day
STAS = ['TA.M17A'] + ['TA.S%02dA' % i for i in range(1, NSTA)]
ef = EventFetcher(SR, ['HHE', 'HHN', 'HHZ'])
for sta in STAS:
    for e in range(len(td)):
        ef.add(sta, 'ev%03d' % e, td[e].reshape(L, NC).T, starttime=1000.0 + 100 * e,
               origintime=990.0 + 100 * e, magnitude=1.0 + (e % 5) * .2)
T = int(hours * 3600 * SR)
con = rs.standard_normal((NC, T)).astype(np.float32)
nplant = max(2, int(50 * hours / 24))
for k in range(nplant):
    e = int(rs.randint(len(td)))
    t = int(rs.randint(0, T - L))
    con[:, t:t + L] += 3.0 * td[e].reshape(L, NC).T / np.std(td[e])
# E2E_RAW=1: the fetcher holds UNFILTERED planes; every chunk is detrended and band-passed on the device
# (detect.py:231 _applyFilter per chunk) before its scan
cf = ContinuousFetcher(SR, ['HHE', 'HHN', 'HHZ'], raw=os.environ.get('E2E_RAW', '0') == '1')
if os.environ.get('E2E_COUNTS', '0') == '1':
    # E2E_COUNTS=1: int32 counts in fragments (one gap per channel), merged / trimmed / filtered on the device
    cnt = np.round(con * 1000.0).astype(np.int32)
    for sta in STAS:
        cf.add_counts(sta, [(c, 0, cnt[c, :T // 2 - 50 * c]) for c in range(NC)] +
                      [(c, T // 2 + 30, cnt[c, T // 2 + 30:]) for c in range(NC)], T, starttime=1.5e9)
else:
  for sta in STAS:
    cf.add(sta, con if sta == STAS[0] else con.copy(), starttime=1.5e9)
if os.environ.get('E2E_PIN', '1') == '1':
    cf.pin()        # page-locked host memory: the H2D copy runs at PCIe rate on a copy stream
t0 = tick('synthetic data (host)', t0)

ctx = HipContext(0)
cl = createCluster(CCreq=0.5, fetch_arg=ef, ctx=ctx)
t0 = tick('createCluster (%d events)' % len(td), t0)
ss = createSubSpace(Pf=1e-10, clust=cl, conDatFetcher=cf)
ss.attachPickTimes(startSeconds=1.0, defaultDuration=30.0)
t0 = tick('createSubSpace + attachPickTimes', t0)
ss.SVD(selectCriteria=4, selectValue=7, threshold=0.3, useSingles=False)
t0 = tick('SVD', t0)
names, kinds, U, off, _, _ = ss._detector_bank('TA.M17A', useSingles=False)
print('detectors: %d, basis vectors: %d, n = %d' % (len(names), U.shape[0], U.shape[1]))
for rep in range(4):
    res = ss.detex(subspaceDB=None, estimateMags=False)
    t0 = tick('detex() pass %d (%d station-days)' % (rep, NSTA), t0)
print('detections: %d of %d planted' % (len(res['ss_df']), nplant))
res = ss.detex(subspaceDB="/tmp/e2e_SubSpace.db", estimateMags=True)
t0 = tick('detex() + magnitudes + SQLite', t0)
ctx.timer_start()
x = torch.from_numpy(np.ascontiguousarray(con.T.reshape(-1))).cuda()
ctx.upload_basis(U, off, NC)
bounds = cf.chunk_bounds('TA.M17A')
t0 = tick('H2D + upload', t0)
