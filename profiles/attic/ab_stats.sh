#!/bin/bash
# A/B of the auxiliary kernels (everything but the scan kernel) from rocprofv3 kernel stats:
#   bash scripts/ab_stats.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python -c "import torch" 2>/dev/null
for V in "$@"; do
  touch detex_amd/csrc/dtx_post.hip detex_amd/csrc/dtx_subspace.hip detex_amd/csrc/dtx_fft.hip detex_amd/csrc/dtx_xcorr.hip
  make -C detex_amd/csrc -j4 EXTRA="-DDTX_FFT_ONLY_VM=10 $V" >/dev/null 2>gpurun_out/ab_build.err || { echo "== EXTRA=$V: BUILD FAILED"; grep -m3 error gpurun_out/ab_build.err; continue; }
  echo "== EXTRA=$V"
  OUT=$ROOT/gpurun_out/ab_stats_prof
  rm -rf $OUT
  (cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT.log 2>&1)
  python3 - <<PY
import csv,glob
for f in glob.glob('$OUT/**/*kernel_stats.csv', recursive=True):
    tot=0
    for r in csv.DictReader(open(f)):
        n=r['Name']; avg=float(r['AverageNs'])/1e3; calls=int(r['Calls'])
        if calls>=20:
            short=n.replace('void ','').replace('(anonymous namespace)::','').split('(')[0][:40]
            print('   %-42s %8.1f us x %d' % (short, avg, calls)); tot+=avg
    print('   sum %.1f us' % tot)
PY
done
