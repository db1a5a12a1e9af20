#!/bin/bash
# A/B variants of the post pass (and of the scan kernel's DS stores) on the GPU box:
#   bash scripts/ab_post.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...
cd ${GRAFT_REPO_ROOT:-.}
for V in "$@"; do
  touch detex_amd/csrc/dtx_fft.hip detex_amd/csrc/dtx_post.hip
  make -C detex_amd/csrc -j4 EXTRA="-DDTX_FFT_ONLY_VM=10 $V" >/dev/null 2>gpurun_out/ab_build.err || { echo "== EXTRA=$V: BUILD FAILED"; grep -m3 error gpurun_out/ab_build.err; continue; }
  echo "== EXTRA=$V"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>gpurun_out/ab_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.4g  ms/step %.3f  kernel_ms %.3f  found %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['planted_found']))"
  bash scripts/kstats.sh 2>/dev/null | grep -E "ds_post|ds_fft_pp"
done
