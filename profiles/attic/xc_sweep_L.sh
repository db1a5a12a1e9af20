#!/bin/bash
# The lag-search kernels of the cross-correlation over event lengths (Nc = 3), on the GPU box:
#   bash scripts/xc_sweep_L.sh [L ...]     (XC_N = events, default 1000; XC_KERNELS = which kernels, default "12 22 8":
#   12 = the half-size workgroup, two per CU (DETEX_AMD_XC12=1), 22 = the 2 x 2 workgroup (DETEX_AMD_XC22=1), 8 = the round-3 kernel)
cd ${GRAFT_REPO_ROOT:-.}
for L in ${@:-1024 1500 2000 2500 3000 3500 4000 5000 5633 6500 9000 13000 20000}; do
  for v in ${XC_KERNELS:-12 22 8}; do
    case $v in
      12) E="DETEX_AMD_XC12=1";;
      22) E="DETEX_AMD_XC12=0 DETEX_AMD_XC22=1";;
      *) E="DETEX_AMD_XC12=0 DETEX_AMD_XC22=0";;
    esac
    env $E python scripts/bench_xcorr.py --events ${XC_N:-1000} --L $L --nc ${XC_NC:-3} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('L=$L Nc=${XC_NC:-3} kernel=$v seconds %.4f kernel_ms %.1f cc_mean %.6f' % (d['seconds'], d['roofline']['kernel_ms'], d['cc_mean']))"
  done
done
