#!/bin/bash
# timing-only ablations of the block-spectra kernel, serial preparation (no side stream):
#   bash scripts/ab_xhat.sh "<EXTRA flags 1>" ...
cd ${GRAFT_REPO_ROOT:-.}
export DETEX_AMD_NO_AUX=1 DTX_BENCH_TIMING_ONLY=1
for V in "$@"; do
  touch detex_amd/csrc/dtx_fft.hip
  make -C detex_amd/csrc -j4 EXTRA="-DDTX_FFT_ONLY_VM=10 $V" >/dev/null 2>gpurun_out/ab_build.err || { echo "== EXTRA=$V: BUILD FAILED"; grep -m3 error gpurun_out/ab_build.err; continue; }
  echo "== EXTRA=$V"
  bash scripts/kstats.sh 2>/dev/null | grep -E "xhat|win_stats|demux|chunk_offset|guard"
done
