# L2 / fabric counters of the scan kernel, one run per setting: bash scripts/pmc_l2.sh "VAR=a" "VAR=b" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  OUT=$ROOT/gpurun_out/prof_l2_$(echo $V | tr -c 'A-Za-z0-9\n' '_')
  mkdir -p $OUT
  export $V
  timeout 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT.log 2>&1
  echo "== $V"
  python3 - <<PY
import csv,glob,collections
for f in glob.glob('$OUT/**/*counter_collection.csv', recursive=True):
    acc=collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "ds_fft_" in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
    for k,v in sorted(acc.items()): print('  %-26s %.4g  (%d launches)' % (k, sum(v)/max(1,len(v)), len(v)))
PY
done
