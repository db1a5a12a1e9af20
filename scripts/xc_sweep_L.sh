#!/bin/bash
# Both lag-search kernels of the cross-correlation over event lengths (N = 1000 events, Nc = 3), on the GPU box:
#   bash scripts/xc_sweep_L.sh [L ...]
cd ${GRAFT_REPO_ROOT:-.}
for L in ${@:-1024 1500 2000 2500 3000 4000 5000 5633 6500 9000 20000}; do
  for v in 1 0; do
    DETEX_AMD_XC22=$v python scripts/bench_xcorr.py --events 1000 --L $L 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('L=$L XC22=$v seconds %.4f kernel_ms %.1f' % (d['seconds'], d['roofline']['kernel_ms']))"
  done
done
