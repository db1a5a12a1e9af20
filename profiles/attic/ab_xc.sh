#!/bin/bash
# A/B variants of the frequency-domain cross-correlation kernel on the GPU box:
#   bash scripts/ab_xc.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...
cd ${GRAFT_REPO_ROOT:-.}
for V in "$@"; do
  touch detex_amd/csrc/dtx_xcfft.hip
  make -C detex_amd/csrc -j4 EXTRA="$V $XC_COMMON" >/dev/null 2>gpurun_out/ab_build.err || { echo "== EXTRA=$V: BUILD FAILED"; grep -m3 error gpurun_out/ab_build.err; continue; }
  echo "== EXTRA=$V"
  python scripts/bench_xcorr.py ${XC_RUN_ARGS:-} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pairs/s %.4g  seconds %.4f  kernel_ms %.1f  cc_mean %.6f' % (d['value'], d['seconds'], d['roofline']['kernel_ms'], d['cc_mean']))"
done
