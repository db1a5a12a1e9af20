#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root:
#   bash scripts/profile.sh <tag>
# Writes rocprofv3 kernel-trace stats and PMC counter CSVs to gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras ${BENCH_ARGS:-}"
# 1. kernel trace + stats (no counters), on the step / warm-up counts the driver's bench run uses, so that the
#    per-kernel average is over warm launches like the live HIP-event figure in the bench line
BENCH_T="python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras ${BENCH_ARGS:-}"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH_T > $OUT/trace.log 2>&1
# 2..4 counters, each in its own pass (FETCH_SIZE and WRITE_SIZE do not fit one pass)
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1
find $OUT -name "*.csv" | head -40
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f; done
tail -2 $OUT/trace.log
