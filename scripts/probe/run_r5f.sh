for i in 1 2 3 4 5; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.readline()); c=o['config']
print('%.4g' % o['value'], 'kernel max %.3f median %.3f' % (c['kernel_ms_max'], c['kernel_ms_median']), 'step median %.3f max %.3f' % (c['step_ms_median'], c['step_ms_max']), 'ms/step %.3f' % o['ms_per_step'])"
done
