cd ${GRAFT_REPO_ROOT:-.}
for V in "$@"; do
  touch detex_amd/csrc/dtx_xcfft.hip
  make -C detex_amd/csrc -j4 EXTRA="$V" >/dev/null 2>gpurun_out/ab_build.err || { echo "== EXTRA=$V: BUILD FAILED"; continue; }
  echo "== EXTRA=$V"
  for L in 1024 3000; do python scripts/bench_xcorr.py --events 2000 --L $L 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  L=$L seconds %.4f kernel_ms %.1f cc_mean %.6f' % (d['seconds'], d['roofline']['kernel_ms'], d['cc_mean']))"; done
done
