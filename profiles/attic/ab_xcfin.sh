#!/bin/bash
# A/B of the cross-correlation finish kernel (rocprofv3 kernel stats of scripts/bench_xcorr.py):
#   bash scripts/ab_xcfin.sh "<EXTRA 1>" ...
cd ${GRAFT_REPO_ROOT:-.}
python -c "import torch" 2>/dev/null
for V in "$@"; do
  touch detex_amd/csrc/dtx_xcorr.hip
  make -C detex_amd/csrc -j4 EXTRA="$V" >/dev/null 2>gpurun_out/ab_build.err || { echo "== EXTRA=$V: BUILD FAILED"; grep -m3 error gpurun_out/ab_build.err; continue; }
  echo "== EXTRA=$V"
  OUT=gpurun_out/ab_xcfin_prof; rm -rf $OUT
  (cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT -- python3 $GRAFT_REPO_ROOT/scripts/bench_xcorr.py > $GRAFT_REPO_ROOT/$OUT.log 2>&1)
  python3 - <<PY
import csv,glob
for f in glob.glob('$OUT/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:3]:
        print('   %-50s %9.1f us x %s' % (r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:50], float(r['AverageNs'])/1e3, r['Calls']))
PY
done
