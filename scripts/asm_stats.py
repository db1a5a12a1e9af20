#!/usr/bin/env python3
"""Instruction mix of the MFMA main loops in a hipcc -save-temps .s file.
usage: asm_stats.py file.s [kernel-name-substring]"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else 'kernel'
for m in re.finditer(r'\.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)', s, re.S):
    if pat in m.group(2):
        print(m.group(2), 'agpr', m.group(1), 'scratch', m.group(3), 'vgpr', m.group(4), 'spill', m.group(5))
for fm in re.finditer(r'^(_Z\S+):\s*; @', s, re.M):
    name = fm.group(1)
    if pat not in name: continue
    i = fm.start(); j = s.index('.Lfunc_end', i)
    blocks = []; cur = None
    for l in s[i:j].split('\n'):
        if re.match(r'^\.LBB\d+_\d+:', l): cur = [l.split(':')[0], []]; blocks.append(cur)
        elif cur is not None: cur[1].append(l.strip())
    for nm, ins in blocks:
        c = lambda p: sum(1 for x in ins if x.startswith(p))
        if c('v_mfma') > 50:
            other = sum(1 for x in ins if x.startswith('v_') and not x.startswith(('v_mfma', 'v_accvgpr', 'v_mov')))
            print(' ', name[-40:], nm, 'mfma', c('v_mfma'), 'ds_read', c('ds_read'), 'acc_mov', c('v_accvgpr'),
                  'scratch', c('scratch_'), 'v_mov', c('v_mov'), 'valu', other, 'glds', c('global_load'),
                  'waitcnt', c('s_waitcnt'), 's_nop', c('s_nop'), 'barrier', c('s_barrier'))
