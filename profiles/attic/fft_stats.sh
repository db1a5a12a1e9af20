#!/bin/bash
# offline look at one kernel instantiation of dtx_fft.hip: registers, spills, instruction mix of its big basic blocks
# (no GPU needed; the Makefile's flags)     bash scripts/fft_stats.sh [mangled-name substring] ["<extra flags>"]
#   default: the configs[1] instantiation ds_fft_pp_kernel<10,false,3>;  e.g. bash scripts/fft_stats.sh ds_fft1_kernelILi14ELb0E
cd /root/repo/detex_amd/csrc && mkdir -p /tmp/fftasm && \
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize $2 -c dtx_fft.hip -o /tmp/fftasm/x.o --save-temps=obj 2>&1 | grep -E "error"
mv dtx_fft-hip* dtx_fft-host* dtx_fft.hip-hip* /tmp/fftasm/ 2>/dev/null
KPAT="${1:-ds_fft_pp_kernelILi10ELb0ELi3E}" python3 - <<'PY'
import os, re
pat=os.environ['KPAT']
s=open('/tmp/fftasm/dtx_fft-hip-amdgcn-amd-amdhsa-gfx950.s').read()
for m in re.finditer(r'\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)', s, re.S):
    if pat in m.group(1): print(m.group(1)[-46:], 'scratch', m.group(2), 'sgpr', m.group(3), 'vgpr', m.group(4), 'spill', m.group(5))
name=[l for l in re.findall(r'^(_ZN\S*):', s, re.M) if pat in l][0]
i=s.index(name+':'); j=s.index('.Lfunc_end',i)
blocks=[]; cur=['entry',[]]; blocks.append(cur)
for l in s[i:j].split('\n'):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: cur=[m.group(1),[]]; blocks.append(cur)
    else: cur[1].append(l.strip())
for nm,ins in blocks:
    c=lambda p: sum(1 for x in ins if x.startswith(p))
    real=[x for x in ins if x and not x.startswith(';') and not x.startswith('.')]
    n=len(real); valu=sum(1 for x in real if x.startswith('v_'))
    if n>80: print(nm,'n',n,'valu',valu,'v_fma',c('v_fma'),'v_mul',c('v_mul_f32'),'add/sub',c('v_add_f32')+c('v_sub_f32'),'pk',c('v_pk_'),'xor',c('v_xor'),'mov',c('v_mov'),'cnd',c('v_cndmask'),'perm',c('v_permlane'),'ds_r',c('ds_read'),'ds_w',c('ds_write'),'gload',c('global_load'),'wait',c('s_waitcnt'),'bar',c('s_barrier'),'scr',c('scratch_'))
PY
