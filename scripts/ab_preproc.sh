#!/bin/bash
# A/B of the pre-processing kernels (scripts/probe/preprocess_rate.py):  bash scripts/ab_preproc.sh "<EXTRA 1>" ...
cd ${GRAFT_REPO_ROOT:-.}
for V in "$@"; do
  touch detex_amd/csrc/dtx_preproc.hip
  make -C detex_amd/csrc -j4 EXTRA="$V" >/dev/null 2>gpurun_out/ab_build.err || { echo "== EXTRA=$V: BUILD FAILED"; grep -m3 error gpurun_out/ab_build.err; continue; }
  echo "== EXTRA=$V"
  python scripts/probe/preprocess_rate.py 2>/dev/null | tail -1
done
