#!/bin/bash
# A/B builds of the block-spectra kernel's loads on the GPU box: bash scripts/probe/ab_xhat.sh "<EXTRA 1>" "<EXTRA 2>" ...
# (quick builds: only the configs[1] instantiation of the scan); per variant the bench step, then the kernel table
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
for V in "$@"; do
  touch detex_amd/csrc/dtx_fft.hip detex_amd/csrc/dtx_xcfft.hip
  make -C detex_amd/csrc -j4 EXTRA="-DDTX_FFT_ONLY_VM=10 $V" >/dev/null 2>gpurun_out/ab_build.err || { echo "== EXTRA=$V: BUILD FAILED"; grep -m3 error gpurun_out/ab_build.err; continue; }
  echo "== EXTRA=$V"
  for rep in 1 2; do
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>gpurun_out/ab_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print('ms/step %.3f  kernel_ms %.3f  step - kernel %.3f  found %s' % (d['ms_per_step'], k, d['ms_per_step']-k, d['config']['planted_found']))"
  done
  rm -rf /tmp/abx; (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/abx -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1)
  f=$(ls /tmp/abx/*/*kernel_stats.csv | head -1)
  python3 - "$f" <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('xhat_kernel','win_stats','demux','ds_post')):
        print('   %-40s calls %4s  avg %8.1f us' % (n[:40], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
