#!/bin/bash
# offline look at xc_fft16_kernel<13>: registers, spills, and the big basic blocks' instruction mix (no GPU needed)
#   bash scripts/xc16_stats.sh "<extra flags>"
cd /root/repo/detex_amd/csrc && mkdir -p /tmp/xcasm && \
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -DDTX_XC16=1 -DDTX_XC16_ONLY=13 $1 -c dtx_xcfft.hip -o /tmp/xcasm/x.o --save-temps=obj 2>&1 | grep -E "error" 
mv /root/repo/detex_amd/csrc/dtx_xcfft-hip* /root/repo/detex_amd/csrc/dtx_xcfft-host* /root/repo/detex_amd/csrc/dtx_xcfft.hip-hip* /tmp/xcasm/ 2>/dev/null
python3 - <<'PY'
import re, collections
s=open('/tmp/xcasm/dtx_xcfft-hip-amdgcn-amd-amdhsa-gfx950.s').read()
for m in re.finditer(r'\.name:\s+(\S*fft16\S*).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)', s, re.S):
    if 'Li13E' in m.group(1): print('scratch', m.group(2), 'sgpr', m.group(3), 'vgpr', m.group(4), 'spill', m.group(5))
name=[l for l in re.findall(r'^(_ZN\S*xc_fft16_kernelILi13E\S*):', s, re.M)][0]
i=s.index(name+':'); j=s.index('.Lfunc_end',i)
blocks=[]; cur=['entry',[]]; blocks.append(cur)
for l in s[i:j].split('\n'):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: cur=[m.group(1),[]]; blocks.append(cur)
    else: cur[1].append(l.strip())
for nm,ins in blocks:
    c=lambda p: sum(1 for x in ins if x.startswith(p))
    n=len([x for x in ins if x and not x.startswith(';') and not x.startswith('.')])
    if n>100: print(nm,'n',n,'scr_st',c('scratch_store'),'scr_ld',c('scratch_load'),'pk_fma',c('v_pk_fma'),'v_fma',c('v_fma'),'ds_read',c('ds_read'),'ds_write',c('ds_write'),'gload',c('global_load'),'s_load',c('s_load'),'waitcnt',c('s_waitcnt'),'barrier',c('s_barrier'),'v_mov',c('v_mov'))
PY
