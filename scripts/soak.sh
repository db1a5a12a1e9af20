#!/bin/bash
# Soak run of the seeded GPU sweeps: every base re-draws the random shapes / data of
# test_gpu_random_sweep, test_gpu_dirty, test_gpu_preproc, test_gpu_ragged, test_gpu_xcorr's lag-search sweep and
# test_gpu_pipeline's trigger-time cases
# (DETEX_TEST_SEED_BASE; 0 = the committed cases).  usage: scripts/soak.sh FIRST LAST  -> gpurun_out/soak.txt
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
first=${1:-1}; last=${2:-4}
: > gpurun_out/soak.txt
for b in $(seq "$first" "$last"); do
  echo "== base $b" >> gpurun_out/soak.txt
  DETEX_TEST_SEED_BASE=$b timeout 900 python -m pytest -q -m gpu -x \
    tests/test_gpu_random_sweep.py tests/test_gpu_dirty.py tests/test_gpu_preproc.py tests/test_gpu_ragged.py \
    "tests/test_gpu_xcorr.py::test_lag_search_random_shapes" "tests/test_gpu_pipeline.py::test_trigger_stats_paths_at_size" 2>&1 | tail -15 >> gpurun_out/soak.txt
done
grep -c passed gpurun_out/soak.txt
tail -5 gpurun_out/soak.txt
