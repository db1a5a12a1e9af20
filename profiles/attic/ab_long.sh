#!/bin/bash
# A/B of the one-group frequency-domain kernel on the cases it still serves: long (segmented) templates and
# 6-channel data.  bash scripts/ab_long.sh "<EXTRA 1>" ...
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for V in "$@"; do
  touch detex_amd/csrc/dtx_fft.hip
  make -C detex_amd/csrc -j4 EXTRA="$V" >/dev/null 2>gpurun_out/ab_build.err || { echo "== EXTRA=$V: BUILD FAILED"; grep -m3 error gpurun_out/ab_build.err; continue; }
  echo "== EXTRA=$V"
  for T in 60 130; do
    python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --template-seconds $T 2>gpurun_out/ab_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('template $T s: ms/step %.3f  kernel_ms %.3f  found %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['planted_found']))"
  done
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --nc 6 --sr 500 --template-seconds 2 --detectors 1 --basis-per-detector 64 --chunks 1 --chunk-seconds 3600 2>gpurun_out/ab_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config5: ms/step %.3f  kernel_ms %.3f  found %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['planted_found']))"
done
