"""Per-unit cadence of SubSpace.detex at 240 station-days (scripts/e2e_240_trace.py's setup): the times at which the
units' results are finished, call by call -- is a slow call slow everywhere, or at a few places?
    python scripts/probe/e2e_unit_times.py [days]"""
import os
import sys
import time

import numpy as np

sys.argv = [sys.argv[0]] + (sys.argv[1:] or ['30'])
os.environ['E2E_CALLS'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detex_amd import detect  # noqa: E402

stamps = []
import gc  # noqa: E402
gc_log = []


def _gc_cb(phase, info):
    if phase == 'start':
        gc_log.append([info['generation'], time.perf_counter()])
    else:
        gc_log[-1].append(time.perf_counter())


gc.callbacks.append(_gc_cb)
if os.environ.get('E2E_NOGC') == '1':
    gc.disable()
_finish = detect.StationScan.finish


def finish(self, *a, **k):
    r = _finish(self, *a, **k)
    stamps.append(time.perf_counter())
    return r


detect.StationScan.finish = finish
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(os.path.dirname(here), 'e2e_240_trace.py')).read()
src = src.replace("for rep in range(3):", "for rep in range(int(os.environ.get('E2E_REPS', '6'))):\n    stamps.clear()\n    gc_log.clear()")
src = src.replace("    print('call %d: %.1f ms, %.3f ms per station-day, %d detections; scan kernel %.3f ms (mean of the contexts\\' last launches)'",
                  "    iv = 1e3 * np.diff(np.array(stamps))\n    print('   garbage collections in the call: %s' % [(g[0], round(1e3 * (g[2] - g[1]), 1)) for g in gc_log if len(g) == 3 and g[2] - g[1] > 2e-3])\n    print('   intervals between finished units (ms): median %.2f  p90 %.2f  max %.2f; the five longest at units %s: %s' % (np.median(iv), np.percentile(iv, 90), iv.max(), np.argsort(iv)[-5:][::-1].tolist(), np.round(np.sort(iv)[-5:][::-1], 1).tolist()))\n    print('call %d: %.1f ms, %.3f ms per station-day, %d detections; scan kernel %.3f ms (mean of the contexts\\' last launches)'")
exec(compile(src, 'e2e_240_trace.py', 'exec'))
