python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_preproc.py tests/test_gpu_ragged.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
python scripts/probe/config5_trace.py 1 | tail -1
for r in 1 0; do
echo "== DETEX_AMD_RAMP=$r"
DETEX_AMD_RAMP=$r python - <<'PY' 2>&1 | tail -3
import bench, json
o = bench.e2e_detex(0)
print({k: (round(v['ms_per_station_day'], 3) if isinstance(v, dict) and 'ms_per_station_day' in v else None) for k, v in o.items()})
o = bench.e2e_detex_240(0)
print({k: (round(v['ms_per_station_day'], 3), [round(x, 3) for x in v['ms_per_station_day_calls']]) for k, v in o.items() if isinstance(v, dict) and 'ms_per_station_day' in v})
PY
done
