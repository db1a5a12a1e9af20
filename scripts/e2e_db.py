"""SubSpace.detex at config-2 size with / without magnitudes and the SQLite file:
[E2E_STATIONS=8] python scripts/e2e_db.py"""
import os
import sys
import time

here = os.path.dirname(os.path.abspath(__file__))
sys.argv = ['scripts/e2e_time.py']
src = open(os.path.join(here, 'e2e_time.py')).read()
head, tail = src.split("for rep in range(4):")
exec(compile(head, 'e2e_head', 'exec'))
for mags, db in [(False, None), (True, None), (True, '/tmp/e2e_x.db'), (False, '/tmp/e2e_x.db')]:
    ss.detex(subspaceDB=db, estimateMags=mags)                          # noqa: F821
    torch.cuda.synchronize()                                            # noqa: F821
    t0 = time.time()
    for _ in range(3):
        ss.detex(subspaceDB=db, estimateMags=mags)                      # noqa: F821
        torch.cuda.synchronize()                                        # noqa: F821
    print('magnitudes %-5s SQLite %-5s: %.1f ms per call' % (mags, bool(db), 1e3 * (time.time() - t0) / 3))
