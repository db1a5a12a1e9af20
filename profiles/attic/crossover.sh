for ts in 0.64 1.28 2.56 5.12 10 20 30 45 60; do
  for e in fft bf16x3; do
    python bench.py --engine $e --template-seconds $ts --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$ts s $e  kernel_ms %.3f  ms/step %.3f' % (d['roofline']['kernel_ms'], d['ms_per_step']))"
  done
done
