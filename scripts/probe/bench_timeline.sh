#!/bin/bash
# kernel trace of the bare bench loop (pipelined scans of one resident station-day): what runs between two scan kernels
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tb
rocprofv3 --kernel-trace --output-format csv -d /tmp/tb -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline > /dev/null 2>&1
f=$(ls /tmp/tb/*/*kernel_trace.csv | head -1)
python3 $GRAFT_REPO_ROOT/scripts/e2e_240_trace.py analyse "$f" 20 > $GRAFT_REPO_ROOT/gpurun_out/bench_timeline.txt 2>&1
