#!/bin/bash
# A/B of run-time switches of the scan kernel on the GPU box (one quick build, then one bench
# per environment setting):  bash scripts/ab_env.sh "<EXTRA flags>" "VAR=a" "VAR=b" ...
cd ${GRAFT_REPO_ROOT:-.}
X="$1"; shift
touch detex_amd/csrc/dtx_fft.hip detex_amd/csrc/dtx_xcfft.hip
make -C detex_amd/csrc -j4 EXTRA="-DDTX_FFT_ONLY_VM=10 $X" >/dev/null 2>gpurun_out/ab_build.err || { echo "== EXTRA=$X: BUILD FAILED"; grep -m3 error gpurun_out/ab_build.err; exit 1; }
for V in "$@"; do
  echo "== $V"
  for rep in 1 2; do
  env $V python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>gpurun_out/ab_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.4g  ms/step %.3f  kernel_ms %.3f  found %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['planted_found']))"
  done
done
