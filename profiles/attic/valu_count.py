"""Static vector-ALU instruction counts of the ping-pong scan kernel, per stretch between
barriers / labels (the kernel is bound by vector-ALU issue: every instruction counts).
python scripts/valu_count.py [extra hipcc flags]"""
import collections
import re
import subprocess
import sys

SRC = 'detex_amd/csrc/dtx_fft.hip'
KERNEL = '_ZN12_GLOBAL__N_116ds_fft_pp_kernelILi10ELb0ELi3EEEvNS_7FftArgsE'
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-DDTX_FFT_ONLY_VM=10',
       '-fno-slp-vectorize', '-S', '--cuda-device-only', SRC, '-o', '/tmp/valu_count.s'] + sys.argv[1:]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
txt = open('/tmp/valu_count.s').read()
body = txt[txt.index(KERNEL + ':'):]
body = body[:body.index('s_endpgm')]
lines = body.split('\n')
meta = re.search(r'\.vgpr_count:\s+(\d+)', txt[txt.index('.name:           ' + KERNEL):][:3000][::-1][::-1])
seg, cur = [], collections.Counter()
start = 0
for i, l in enumerate(lines):
    t = l.strip()
    if t.startswith('s_barrier') or re.match(r'^\.LBB\d+_\d+:', t):
        seg.append((start, i, t.split()[0], cur))
        cur, start = collections.Counter(), i
    else:
        op = t.split()[0] if t else ''
        if op.startswith('v_'):
            cur['valu'] += 1
            if op.startswith('v_mov'):
                cur['mov'] += 1
        elif op.startswith('ds_'):
            cur['lds'] += 1
        elif op.startswith(('global_', 'scratch_')):
            cur['mem'] += 1
tot = 0
for a, b, mark, c in seg:
    tot += c['valu']
    if c['valu'] + c['lds'] + c['mem'] > 8:
        print('%5d-%5d  valu %4d (mov %3d)  lds %3d  mem %3d' % (a, b, c['valu'], c['mov'], c['lds'], c['mem']))
print('total valu', tot)
m = re.search(KERNEL + r'.*?\.vgpr_count:\s+(\d+)', txt, re.S)
m2 = re.search(r'\.private_segment_fixed_size:\s+(\d+)\s+\.sgpr_count:\s+\d+\s+\.sgpr_spill_count:\s+\d+\s+\.symbol:\s+' + KERNEL, txt)
print('vgprs', m.group(1) if m else '?', ' scratch', m2.group(1) if m2 else '?')
