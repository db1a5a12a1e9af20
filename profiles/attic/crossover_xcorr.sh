for L in 256 512 768 1024 2048; do
  for e in fft bf16x3; do
    python scripts/bench_xcorr.py --events 600 --L $L --engine $e --steps 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('L=$L $e  %.3f ms  (%s)' % (1e3*d['seconds'], d['roofline'].get('kernel','')[:20]))"
  done
done
