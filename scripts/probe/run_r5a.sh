python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -3
python - <<'PY' 2>&1 | tail -12
import bench, json, torch
print(json.dumps(bench.e2e_createcluster(0, cpu=False))[:1500])
print(json.dumps(bench.e2e_fas(0))[:900])
PY
python scripts/probe/config5_trace.py 1; python scripts/probe/config5_trace.py 24
cd /tmp && export TMPDIR=/tmp
for h in 1 24; do
rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $GRAFT_REPO_ROOT/scripts/probe/config5_trace.py $h > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/ks/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print('%-60s calls %4s avg %10.1f us  %5s%%' % (r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:60], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
done
