#!/bin/bash
# A/B variants of the post pass alone (scripts/probe/read_bw.bin times dtx_ds_post on an 886 MB tensor):
#   bash scripts/ab_post2.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...
cd ${GRAFT_REPO_ROOT:-.}
for V in "$@"; do
  touch detex_amd/csrc/dtx_post.hip
  make -C detex_amd/csrc -j4 EXTRA="$V" >/dev/null 2>gpurun_out/ab_build.err || { echo "== EXTRA=$V: BUILD FAILED"; grep -m3 error gpurun_out/ab_build.err; continue; }
  echo "== EXTRA=$V"
  timeout 120 scripts/probe/read_bw.bin 2>&1 | grep "dtx_ds_post data=noise"
done
