#!/bin/bash
# host-side launch times against device start times in the steady state of SubSpace.detex (is the host late?)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/t240h
rocprofv3 --kernel-trace --hip-runtime-trace --memory-copy-trace --output-format csv -d /tmp/t240h -- python3 $GRAFT_REPO_ROOT/scripts/e2e_240_trace.py 6 > $GRAFT_REPO_ROOT/gpurun_out/t240h.txt 2>&1
ls /tmp/t240h/*/ >> $GRAFT_REPO_ROOT/gpurun_out/t240h.txt
python3 - >> $GRAFT_REPO_ROOT/gpurun_out/t240h.txt 2>&1 <<'PY'
import csv, glob
d = glob.glob('/tmp/t240h/*/')[0]
kt = list(csv.DictReader(open(glob.glob(d + '*kernel_trace.csv')[0])))
api = list(csv.DictReader(open(glob.glob(d + '*hip_api_trace.csv')[0])))
mc = glob.glob(d + '*memory_copy_trace.csv')
print('api columns:', ', '.join(api[0].keys()))
by = {}
for r in api:
    by.setdefault(r['Correlation_Id'], r)
scans = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Correlation_Id']) for r in kt if 'ds_fft' in r['Kernel_Name'])
lo, hi = scans[-41][0], scans[-1][0]
stat = {}
for r in kt:
    st = int(r['Start_Timestamp'])
    if not (lo <= st < hi): continue
    a = by.get(r['Correlation_Id'])
    if a is None: continue
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:34]
    stat.setdefault(nm, []).append((st - int(a['Start_Timestamp'])) / 1e3)
print('device start minus host launch call, us (mean / min / max), steady state, 40 units:')
for nm, v in sorted(stat.items(), key=lambda kv: -len(kv[1])):
    print('   %-36s n %4d  mean %9.1f  min %9.1f  max %9.1f' % (nm, len(v), sum(v) / len(v), min(v), max(v)))
# one unit: host calls in order between two scan launches
a0 = int(by[scans[-20][2]]['Start_Timestamp']); a1 = int(by[scans[-19][2]]['Start_Timestamp'])
print('host API calls between two consecutive scan-kernel launches (%.1f us apart):' % ((a1 - a0) / 1e3))
for r in sorted(api, key=lambda r: int(r['Start_Timestamp'])):
    st = int(r['Start_Timestamp'])
    if a0 <= st <= a1:
        print('   %9.1f us  %7.1f us  %s' % ((st - a0) / 1e3, (int(r['End_Timestamp']) - st) / 1e3, r['Function']))
if mc:
    rows = list(csv.DictReader(open(mc[0])))
    print('memory copies in the steady state:', ', '.join(rows[0].keys()))
    tot = {}
    for r in rows:
        st = int(r['Start_Timestamp'])
        if lo <= st < hi:
            k = (r.get('Direction', '?'), 'big' if int(r.get('Bytes', r.get('Size', 0)) or 0) > 1 << 20 else 'small')
            tot.setdefault(k, []).append((int(r['End_Timestamp']) - st) / 1e3)
    for k, v in tot.items():
        print('   %-28s n %4d  mean %8.1f us' % (k, len(v), sum(v) / len(v)))
PY
