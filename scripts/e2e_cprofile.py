"""Host-side profile (cProfile) of SubSpace.detex over 8 station-days: where the Python time goes.
[E2E_STATIONS=8] python scripts/e2e_cprofile.py"""
import cProfile
import os
import pstats
import sys

os.environ.setdefault('E2E_STATIONS', '8')
here = os.path.dirname(os.path.abspath(__file__))
sys.argv = ['scripts/e2e_time.py']
src = open(os.path.join(here, 'e2e_time.py')).read()
head, tail = src.split("for rep in range(4):")
exec(compile(head, 'e2e_head', 'exec'))
for _ in range(2):
    ss.detex(subspaceDB=None, estimateMags=True)                       # noqa: F821
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    ss.detex(subspaceDB=None, estimateMags=True)                       # noqa: F821
torch.cuda.synchronize()                                               # noqa: F821
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)
st.sort_stats('tottime').print_stats(18)
