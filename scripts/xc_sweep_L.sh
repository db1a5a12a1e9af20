#!/bin/bash
# The lag-search kernels of the cross-correlation over event lengths (Nc = 3), on the GPU box:
#   bash scripts/xc_sweep_L.sh [L ...]        (XC_N = events, default 1000; XC_KERNELS = which of DETEX_AMD_XC22, default "1 0")
cd ${GRAFT_REPO_ROOT:-.}
for L in ${@:-1024 1500 2000 2500 3000 3500 4000 5000 5633 6500 9000 20000}; do
  for v in ${XC_KERNELS:-1 0}; do
    DETEX_AMD_XC22=$v python scripts/bench_xcorr.py --events ${XC_N:-1000} --L $L 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('L=$L XC22=$v seconds %.4f kernel_ms %.1f cc_mean %.6f' % (d['seconds'], d['roofline']['kernel_ms'], d['cc_mean']))"
  done
done
