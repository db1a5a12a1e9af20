#!/bin/bash
# A/B kernel variants on the GPU box: bash scripts/ab.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...
cd ${GRAFT_REPO_ROOT:-.}
for V in "$@"; do
  make -C detex_amd/csrc clean >/dev/null; make -C detex_amd/csrc -j8 EXTRA="$V" >/dev/null 2>&1 || echo BUILD FAILED
  echo "== EXTRA=$V"
  [ -z "$NOTEST" ] && python -m pytest tests -m gpu -q -x 2>&1 | tail -1
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done
