#!/bin/bash
# Soak run of the seeded cases of test_gpu_engines / test_gpu_subspace / test_gpu_xcorr and of test_gpu_pipeline's
# workflow (DETEX_TEST_SEED_BASE re-draws their data; fixtures and golden vectors stay; expectations that depend on
# the draw -- how many planted events trigger -- hold for the committed draw only).  usage: scripts/soak2.sh FIRST LAST  -> gpurun_out/soak2.txt
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
first=${1:-1}; last=${2:-4}
: > gpurun_out/soak2.txt
for b in $(seq "$first" "$last"); do
  echo "== base $b" >> gpurun_out/soak2.txt
  DETEX_TEST_SEED_BASE=$b timeout 900 python -m pytest -q -m gpu tests/test_gpu_engines.py tests/test_gpu_subspace.py \
    tests/test_gpu_xcorr.py tests/test_gpu_pipeline.py 2>&1 | grep -E "^FAILED|^E  .*(assert|Error)|passed|failed" | head -40 >> gpurun_out/soak2.txt
done
grep -c "passed" gpurun_out/soak2.txt
grep -E "FAILED" gpurun_out/soak2.txt | sort | uniq -c | sort -rn | head -20
