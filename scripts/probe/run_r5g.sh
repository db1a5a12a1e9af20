#!/bin/bash
# steady state of SubSpace.detex over day units under a kernel trace (scripts/e2e_240_trace.py) -> gpurun_out/t240.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/t240
rocprofv3 --kernel-trace --output-format csv -d /tmp/t240 -- python3 $GRAFT_REPO_ROOT/scripts/e2e_240_trace.py 6 > $GRAFT_REPO_ROOT/gpurun_out/t240.txt 2>&1
f=$(ls /tmp/t240/*/*kernel_trace.csv | head -1)
python3 $GRAFT_REPO_ROOT/scripts/e2e_240_trace.py analyse "$f" 40 >> $GRAFT_REPO_ROOT/gpurun_out/t240.txt 2>&1
# the same three calls without the profiler
python3 $GRAFT_REPO_ROOT/scripts/e2e_240_trace.py 6 2>&1 | grep "^call" >> $GRAFT_REPO_ROOT/gpurun_out/t240.txt
