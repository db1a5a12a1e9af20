"""cProfile of SubSpace.detex() at config-2 size (after one warm call): where the host time of
a station-day goes.  Usage: [E2E_STATIONS=8] python scripts/e2e_profile.py [mags]
(the default path: per-station contexts, two stations in flight)"""
import cProfile
import os
import pstats
import sys

mags = len(sys.argv) > 1 and sys.argv[1] == 'mags'
sys.argv = ['scripts/e2e_time.py']
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'e2e_time.py')).read()
head, tail = src.split("for rep in range(4):")
exec(compile(head, 'e2e_head', 'exec'))
res = ss.detex(subspaceDB=None, estimateMags=mags)                   # noqa: F821  (warm)
torch.cuda.synchronize()                                             # noqa: F821
pr = cProfile.Profile()
pr.enable()
for rep in range(5):
    res = ss.detex(subspaceDB=None, estimateMags=mags)               # noqa: F821
    torch.cuda.synchronize()                                         # noqa: F821
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(55)
