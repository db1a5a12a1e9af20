import cProfile, pstats, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from detex_amd import EventFetcher, createCluster, synth
from detex_amd.hip import HipContext
SR, NC, L, NFAM, NPER = 100.0, 3, 3300, 25, 8
td = synth.event_families(1, NFAM, NPER, NC, L, maxshift=20, snr=8.0)
ef = EventFetcher(SR, ['HHE', 'HHN', 'HHZ'])
for e in range(len(td)):
    ef.add('TA.M17A', 'ev%03d' % e, td[e].reshape(L, NC).T, starttime=1000.0 + 100 * e, origintime=990.0 + 100 * e, magnitude=1.0)
ctx = HipContext(0)
cl = createCluster(CCreq=0.5, fetch_arg=ef, ctx=ctx)
pr = cProfile.Profile(); pr.enable()
for _ in range(3): cl = createCluster(CCreq=0.5, fetch_arg=ef, ctx=ctx)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
