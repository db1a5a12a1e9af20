#!/bin/bash
# rocprofv3 evidence for the cross-correlation kernels at configs[2] (2000 events x 39 000 samples).
# Run on the GPU box (through gpurun) from the repo root:
#   bash scripts/profile_xc.sh <tag>
# Kernel trace + stats in one pass, each counter group in a pass of its own (gpurun refuses --pmc
# together with the trace domains).  Summaries: scripts/summarize_xc.py <tag>.
set -u
TAG=${1:-x4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RUN="python3 $ROOT/scripts/bench_xcorr.py --steps 2 ${XC_ARGS:-}"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $RUN > $OUT/trace.log 2>&1
RUN1="python3 $ROOT/scripts/bench_xcorr.py --steps 1 ${XC_ARGS:-}"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $RUN1 > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $RUN1 > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/pmc_tcc -- $RUN1 > $OUT/pmc_tcc.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- $RUN1 > $OUT/pmc_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $RUN1 > $OUT/pmc_sq2.log 2>&1
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f; done
tail -2 $OUT/trace.log
