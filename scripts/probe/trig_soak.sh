#!/bin/bash
# re-draws of the trigger-time kernel's parity tests (tests/test_gpu_pipeline.py -k trigger_stats) under seed bases $1..$2
cd ${GRAFT_REPO_ROOT:-.}
for b in $(seq $1 $2); do
  r=$(DETEX_TEST_SEED_BASE=$b timeout 600 python -m pytest -q -m gpu -x tests/test_gpu_pipeline.py -k "trigger_stats or in_flight or day_units" 2>&1 | tail -1)
  echo "seed base $b: $r"
done > gpurun_out/trig_soak.txt 2>&1
tail -3 gpurun_out/trig_soak.txt
