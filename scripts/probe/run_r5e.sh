for r in 0 1 0 1 0 1; do
echo "== DETEX_AMD_RAMP=$r"
DETEX_AMD_RAMP=$r python - <<'PY' 2>&1 | tail -1
import bench
o = bench.e2e_detex(0)
print({k: round(v['ms_per_station_day'], 3) for k, v in o.items() if isinstance(v, dict) and 'ms_per_station_day' in v})
PY
done
