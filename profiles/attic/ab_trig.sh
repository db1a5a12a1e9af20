#!/bin/bash
# A/B builds of the trigger-time kernel on the GPU box:  bash scripts/ab_trig.sh "<EXTRA flags 1>" ...
cd ${GRAFT_REPO_ROOT:-.}
for V in "$@"; do
  touch detex_amd/csrc/dtx_trig.hip
  make -C detex_amd/csrc -j4 EXTRA="$V" >/dev/null 2>gpurun_out/ab_build.err || { echo "== EXTRA=$V: BUILD FAILED"; grep -m3 error gpurun_out/ab_build.err; continue; }
  echo "== EXTRA=$V"
  python scripts/probe/trig_time.py 2>/dev/null | tail -2
done
