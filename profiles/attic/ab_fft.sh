#!/bin/bash
# A/B variants of the frequency-domain engine on the GPU box:
#   bash scripts/ab_fft.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...
cd ${GRAFT_REPO_ROOT:-.}
for V in "$@"; do
  touch detex_amd/csrc/dtx_fft.hip
  make -C detex_amd/csrc EXTRA="$V" >/dev/null 2>&1 || echo BUILD FAILED
  echo "== EXTRA=$V"
  [ -z "$NOTEST" ] && python -m pytest tests/test_gpu_engines.py -q -x -k fft 2>&1 | tail -1
  python bench.py --engine fft --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.4g  ms/step %.3f  kernel_ms %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
done
