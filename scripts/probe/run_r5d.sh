cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr8 && E2E_STATIONS=8 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr8 -- python3 $GRAFT_REPO_ROOT/scripts/e2e_time.py > /tmp/tr8.log 2>&1
tail -8 /tmp/tr8.log
f=$(find /tmp/tr8 -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/scripts/e2e_gaps.py $f 10
