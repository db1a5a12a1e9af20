"""Average the s_memtime stamps of a -DDTX_PP_STAMPS build of the scan kernel
(DTX_PP_STAMP_FILE=<file> python bench.py ...): python scripts/pp_stamps.py <file> [first_round last_round]"""
import sys

import numpy as np

R, W = 24, 56
a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(W, 2, R, 16)
r0 = int(sys.argv[2]) if len(sys.argv) > 2 else 6
r1 = int(sys.argv[3]) if len(sys.argv) > 3 else 22
FRONT = ['pass 1 (registers)', 'wait barrier 1', 'exchange 1 (write, read)', 'pass 2 + write', 'exchange 2 read',
         'pass 3 + write', 'wait barrier 2', 'cross-wave reads']
MAC = ['prefetch issue + wait barrier 1', 'tail: radix 8 x 2', 'accumulate', 'statistic (div)', 'stores + products',
       'pass 1 + wait barrier 2']
for half in (0, 1):
    for role, names in (('front', FRONT), ('mac', MAC)):
        rows = []
        for w in range(W):
            for r in range(r0, r1):
                s = a[w, half, r]
                if s[0] == 0:
                    continue
                if ((r - half) & 1) != (role == 'front'):
                    continue
                n = len(names) + 1
                if np.any(s[:n] == 0):
                    continue
                rows.append(np.diff(s[:n]))
        if not rows:
            continue
        m = np.mean(rows, axis=0)
        print('half %d, %s role: %d samples, total %.0f ticks' % (half, role, len(rows), m.sum()))
        for nm, v in zip(names, m):
            print('   %-30s %7.0f' % (nm, v))
# round length: start of round r+1 minus start of round r
d = []
for w in range(W):
    for half in (0, 1):
        s = a[w, half, r0:r1 + 1, 0]
        if np.all(s > 0):
            d.append(np.diff(s))
if d:
    d = np.array(d)
    print('round length: mean %.0f ticks (rounds %d..%d, %d rows)' % (d.mean(), r0, r1, len(d)))
