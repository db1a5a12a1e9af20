cd ${GRAFT_REPO_ROOT:-.}
for V in "" "-DDTX_FFT_SEG_L=4097"; do
  make -C detex_amd/csrc clean >/dev/null; make -C detex_amd/csrc -j8 EXTRA="$V" >/dev/null 2>&1 || echo BUILD FAILED
  echo "== $V"
  for ts in 42 46 50 55 60; do
    python bench.py --template-seconds $ts --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$ts s kernel_ms %.3f ms/step %.3f' % (d['roofline']['kernel_ms'], d['ms_per_step']))"
  done
done
