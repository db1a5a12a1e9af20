#!/bin/bash
# Diagnostic: per-phase s_memtime stamps of the ping-pong scan kernel at configs[1].
# bash scripts/stamps.sh <tag> [EXTRA flags]
cd ${GRAFT_REPO_ROOT:-.}
touch detex_amd/csrc/dtx_fft.hip
make -C detex_amd/csrc -j4 EXTRA="-DDTX_FFT_ONLY_VM=10 -DDTX_PP_STAMPS $2" >/dev/null 2>gpurun_out/ab_build.err || { echo "BUILD FAILED"; grep -m3 error gpurun_out/ab_build.err; exit 1; }
DTX_BENCH_TIMING_ONLY=1 DTX_PP_STAMP_FILE=gpurun_out/stamps_$1.bin python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>gpurun_out/ab_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_ms %.3f (with stamps)  found %s' % (d['roofline']['kernel_ms'], d['config']['planted_found']))"
python scripts/pp_stamps.py gpurun_out/stamps_$1.bin | tee gpurun_out/stamps_$1.txt
