#!/bin/bash
# kernel trace of SubSpace.detex over day units with an UNFILTERED fetcher (detrend + band-pass on the device per chunk)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/t240r
rocprofv3 --kernel-trace --output-format csv -d /tmp/t240r -- python3 $GRAFT_REPO_ROOT/scripts/e2e_240_trace.py 12 raw > /tmp/t240r.log 2>&1
grep "call" /tmp/t240r.log > $GRAFT_REPO_ROOT/gpurun_out/e2e_timeline_raw.txt
f=$(ls /tmp/t240r/*/*kernel_trace.csv | head -1)
python3 $GRAFT_REPO_ROOT/scripts/e2e_240_trace.py analyse "$f" 80 >> $GRAFT_REPO_ROOT/gpurun_out/e2e_timeline_raw.txt 2>&1
