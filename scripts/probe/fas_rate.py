"""SubSpace.getFAS (fas.py:72-86 of the reference: the null distribution of the statistic over random chunks) at the
e2e workload: where its time goes."""
import cProfile
import os
import pstats
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
os.environ.setdefault('E2E_STATIONS', '1')
sys.argv = ['scripts/e2e_time.py']
src = open(os.path.join(root, 'scripts', 'e2e_time.py')).read()
head, tail = src.split("for rep in range(4):")
exec(compile(head, 'e2e_head', 'exec'))
ss.getFAS(6, seed=1)                                                 # noqa: F821  (warm)
t0 = time.perf_counter()
ss.getFAS(24, seed=2)                                                # noqa: F821
torch.cuda.synchronize()                                             # noqa: F821
print('getFAS(24 chunks of a station-day): %.1f ms' % (1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile()
pr.enable()
ss.getFAS(24, seed=3)                                                # noqa: F821
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(10)
