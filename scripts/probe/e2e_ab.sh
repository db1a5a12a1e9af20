#!/bin/bash
# A/B of an environment switch on the e2e legs: bash scripts/probe/e2e_ab.sh VAR a b [rounds]
cd ${GRAFT_REPO_ROOT:-.}
VAR=$1; A=$2; B=$3; N=${4:-2}
for i in $(seq $N); do
  for V in $A $B; do
    echo "== $VAR=$V"
    env $VAR=$V python scripts/e2e_240.py 2>/dev/null | python -c "
import sys, json
d = json.load(sys.stdin)
print('  e2e_240 ms per station-day: no_db %.3f  sqlite %.3f  raw_fetcher %.3f' % tuple(d[k]['ms_per_station_day'] for k in ('no_db', 'sqlite', 'raw_fetcher')))"
  done
done
