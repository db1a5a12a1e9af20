#!/bin/bash
# HBM-side fetch traffic (FETCH_SIZE, KiB as reported; x2 on gfx950) of the scan kernel per build variant:
#   bash scripts/ab_traffic.sh "<EXTRA flags 1>" ...
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$(pwd)
for V in "$@"; do
  touch detex_amd/csrc/dtx_fft.hip
  make -C detex_amd/csrc -j4 EXTRA="-DDTX_FFT_ONLY_VM=10 $V" >/dev/null 2>gpurun_out/ab_build.err || { echo "== EXTRA=$V: BUILD FAILED"; grep -m3 error gpurun_out/ab_build.err; continue; }
  OUT=$ROOT/gpurun_out/traffic_tmp; rm -rf $OUT
  (cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1)
  python3 - "$V" <<PY
import csv,glob,sys
v=[float(r['Counter_Value']) for f in glob.glob('$OUT/**/*counter_collection.csv', recursive=True) for r in csv.DictReader(open(f)) if 'ds_fft_' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE']
print('== EXTRA=%s  FETCH_SIZE x2 = %.2f GB per launch (%d launches)' % (sys.argv[1], 2*sum(v)/len(v)*1024/1e9, len(v)))
PY
done
