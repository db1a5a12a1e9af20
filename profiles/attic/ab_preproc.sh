#!/bin/bash
# A/B of the pre-processing kernels (scripts/probe/preprocess_rate.py, kernel table under rocprofv3):
#   bash scripts/ab_preproc.sh "<EXTRA 1>" ...
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
for V in "$@"; do
  touch detex_amd/csrc/dtx_preproc.hip
  make -C detex_amd/csrc -j4 EXTRA="$V" >/dev/null 2>gpurun_out/ab_build.err || { echo "== EXTRA=$V: BUILD FAILED"; grep -m3 error gpurun_out/ab_build.err; continue; }
  echo "== EXTRA=$V"
  python scripts/probe/preprocess_rate.py 2>/dev/null | tail -1
  rm -rf /tmp/pp_prof; (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_prof -- python3 $ROOT/scripts/probe/preprocess_rate.py > /dev/null 2>&1)
  python3 - <<PY
import csv, glob
for f in glob.glob('/tmp/pp_prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        if 'pp_' in r['Name']: print('   %-62s %8.1f us x %s' % (r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:62], float(r['AverageNs']) / 1e3, r['Calls']))
PY
done
