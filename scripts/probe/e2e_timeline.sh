#!/bin/bash
# kernel trace of SubSpace.detex over day units (scripts/e2e_240_trace.py), with and without the magnitude
# estimates: period, GPU busy time, the kernels' shares and what runs between two scan kernels
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  rm -rf /tmp/t240
  E2E_MAGS=$m rocprofv3 --kernel-trace --output-format csv -d /tmp/t240 -- python3 $GRAFT_REPO_ROOT/scripts/e2e_240_trace.py 30 > /tmp/t240.log 2>&1
  grep "call" /tmp/t240.log
  f=$(ls /tmp/t240/*/*kernel_trace.csv | head -1)
  echo "==== estimateMags=$m"
  python3 $GRAFT_REPO_ROOT/scripts/e2e_240_trace.py analyse "$f" 200
done > $GRAFT_REPO_ROOT/gpurun_out/e2e_timeline_ab.txt 2>&1
