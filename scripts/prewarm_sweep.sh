#!/bin/bash
# Verdict r4 item 2: the headline on ONE fresh lease under --prewarm 0 / 0.5 (fixed, round 4) / 2 / 5 (cap, settled
# criterion); each run is its own process = a GPU that idled while the previous one tore down.  First thing on the box.
mkdir -p gpurun_out
for pw in "0 0" "0.5 0.5" "2 0.5" "5 0.5"; do
  set -- $pw
  python bench.py --steps 20 --warmup 5 --prewarm $1 --prewarm-min $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.readline()); c=o['config']
print('prewarm cap $1 min $2: value %.4g ms/step %.3f kernel min/med/max %.3f %.3f %.3f step med %.3f prewarm %s' % (o['value'], o['ms_per_step'], c['kernel_ms_min'], c['kernel_ms_median'], c['kernel_ms_max'], c['step_ms_median'], json.dumps(c['prewarm'])))"
done
