#!/bin/bash
# A/B variants of the frequency-domain scan kernel on the GPU box (quick builds: only the
# configs[1] instantiation):  bash scripts/ab_pp.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...
cd ${GRAFT_REPO_ROOT:-.}
for V in "$@"; do
  touch detex_amd/csrc/dtx_fft.hip detex_amd/csrc/dtx_xcfft.hip
  make -C detex_amd/csrc -j4 EXTRA="-DDTX_FFT_ONLY_VM=10 $V" >/dev/null 2>gpurun_out/ab_build.err || { echo "== EXTRA=$V: BUILD FAILED"; grep -m3 error gpurun_out/ab_build.err; continue; }
  echo "== EXTRA=$V"
  for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>gpurun_out/ab_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.4g  ms/step %.3f  kernel_ms %.3f  found %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['planted_found']))"
  done
done
