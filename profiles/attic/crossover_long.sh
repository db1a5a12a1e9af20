for ts in 61.4 62 90 130 160; do
  for e in fft bf16x3; do
    python bench.py --engine $e --template-seconds $ts --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$ts s $e  kernel_ms %.3f  ms/step %.3f  found %s' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['config']['planted_found']))"
  done
done
