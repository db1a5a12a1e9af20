ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${1:-extra}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --engine fft"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/a -- $BENCH > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/b -- $BENCH > $OUT/b.log 2>&1
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/c -- $BENCH > $OUT/c.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in 'abc':
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv'%d, recursive=True):
        acc=collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "ds_fft_" in row['Kernel_Name']:
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
        agg=collections.defaultdict(float); n=collections.Counter()
        # sum over dispatch dims per dispatch: values per (dispatch) may be split; just average per launch by summing / launches
        for k,v in acc.items(): print(d,k,sum(v)/max(1,len(v)), len(v))
PY
tail -3 $OUT/c.log
