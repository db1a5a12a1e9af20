#!/bin/bash
# Round 6, final evidence on one lease: GPU suite, default bench line, rocprofv3 stats + counters of the scan
# (configs[1]) and of configs[4]'s kernel, the cross-correlation's, the product path's timeline.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests/ -q -m gpu 2>&1 | tail -3 > gpurun_out/r6d_tests.txt
python bench.py > gpurun_out/r6d_bench.json 2> gpurun_out/r6d_bench.err
bash scripts/profile.sh r6d > gpurun_out/r6d_profile.log 2>&1
BENCH_ARGS="--nc 6 --sr 500 --template-seconds 2 --detectors 1 --basis-per-detector 64 --chunks 1 --chunk-seconds 3600" bash scripts/profile.sh r6c5 > gpurun_out/r6c5_profile.log 2>&1
bash scripts/profile_xc.sh x16 > gpurun_out/x16_profile.log 2>&1
python scripts/summarize_profile.py r6d ds_fft_pp_kernel > /dev/null 2>&1
python scripts/summarize_profile.py r6c5 ds_fft_kernel > /dev/null 2>&1
python scripts/summarize_xc.py x16 > /dev/null 2>&1
cp profiles/r6d_bench_kernel_stats.csv profiles/r6d_bench_pmc_ds_fft_pp_kernel.json profiles/r6c5_bench_kernel_stats.csv profiles/r6c5_bench_pmc_ds_fft_kernel.json profiles/x16_* gpurun_out/ 2>/dev/null
bash scripts/probe/run_r5g.sh > /dev/null 2>&1
cat gpurun_out/r6d_tests.txt
python - <<PY
import json
d=json.load(open("gpurun_out/r6d_bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"].get("traffic"))
print({k:v for k,v in d["config"].items() if k.endswith("_step") or "over" in k or "e2e" in k or "xcorr" in k})
print("config5", d["config5"]["ms_per_hour_of_data"], d["config5"]["roofline"]["kernel"], d["config5"]["roofline"]["traffic"])
print("createCluster", d["e2e_createCluster_2000"]["seconds"], "xc", d["xcorr_full"]["roofline"]["kernel"], d["xcorr_full"]["short_variant"]["roofline"]["kernel"])
PY
ls gpurun_out | grep -E "r6d|r6c5|x16" | head -20
