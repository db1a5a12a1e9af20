#!/bin/bash
# per-kernel durations of the default bench run (rocprofv3 --kernel-trace --stats), top rows
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/ks/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print('%-48s calls %4s avg %10.1f us  %5s%%' % (r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:48], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
