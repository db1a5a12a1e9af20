python scripts/probe/config5_trace.py 1 | tail -1
for n in 1 8; do
echo "=== E2E_STATIONS=$n"
E2E_STATIONS=$n python scripts/e2e_cprofile.py 2>&1 | grep -v "^$" | head -75
done
python - <<'PY' 2>&1 | tail -4
import bench, json
print(json.dumps(bench.e2e_createcluster(0, cpu=False))[:1400])
PY
