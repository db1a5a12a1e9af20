"""Ad-hoc wider sweep: reuse the committed sweep tests with many more seeds."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_gpu_random_sweep as t
from detex_amd.hip import HipContext
ctx = HipContext(0)
bad = 0
for seed in range(24, 24 + int(sys.argv[1])):
    try:
        for eng in ('mfma', 'fft'):
            t.test_random_scan(ctx, seed, eng)
    except Exception as e:
        bad += 1
        print('SCAN seed', seed, repr(e)[:300])
        for tag, c in (('same ctx', ctx), ('new ctx', HipContext(0))):
            try:
                t.test_random_scan(c, seed, 'fft')
                print('   rerun', tag, 'ok')
            except Exception as e2:
                print('   rerun', tag, repr(e2)[:300])
for seed in range(8, 8 + int(sys.argv[2])):
    try:
        t.test_random_xcorr(ctx, seed)
    except Exception as e:
        bad += 1
        print('XCORR seed', seed, repr(e)[:300])
print('failures:', bad)
