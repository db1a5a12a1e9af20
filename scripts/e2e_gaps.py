"""GPU idle gaps of one detex() pass from a rocprofv3 kernel trace of scripts/e2e_time.py:
python scripts/e2e_gaps.py <kernel_trace.csv> [stations]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nsta = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']),
             r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:38]) for r in rows)
pp = [e for e in ev if 'pp_copy' in e[2]]
# pass 2 of the timing loop: stations 2*nsta .. 3*nsta - 1 (pass 0 is the warm one)
t0, t1 = pp[2 * nsta][0], pp[3 * nsta][0]
win = [e for e in ev if e[0] >= t0 and e[0] < t1]
busy, end, gaps = 0, win[0][0], []
for s, e, n in win:
    if s > end:
        gaps.append((s - end, n, (s - t0) / 1e6))
        end = s
    if e > end:
        busy += e - end
        end = e
print('pass: %.2f ms, busy %.2f ms, %d kernels' % ((t1 - t0) / 1e6, busy / 1e6, len(win)))
tot = {}
for s, e, n in win:
    tot[n] = tot.get(n, 0) + (e - s)
for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:8]:
    print('   %-40s %7.2f ms' % (n, v / 1e6))
print('largest gaps (us, before kernel, at ms):')
for g, n, at in sorted(gaps, reverse=True)[:14]:
    print('   %6d  %-34s %7.2f' % (g / 1e3, n, at))
