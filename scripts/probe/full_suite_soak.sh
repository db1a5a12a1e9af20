#!/bin/bash
# the WHOLE GPU suite under re-drawn seeds (every file that reads DETEX_TEST_SEED_BASE): usage full_suite_soak.sh FIRST LAST
cd ${GRAFT_REPO_ROOT:-.}
: > gpurun_out/full_soak.txt
for b in $(seq $1 $2); do
  r=$(DETEX_TEST_SEED_BASE=$b timeout 900 python -m pytest tests/ -q -m gpu 2>&1 | tail -1)
  echo "seed base $b: $r" >> gpurun_out/full_soak.txt
done
cat gpurun_out/full_soak.txt
