#!/bin/bash
# offline look at one kernel instantiation of dtx_xcfft.hip: registers, spills, instruction mix of its big basic
# blocks (no GPU needed; the Makefile's flags)      bash scripts/xc_stats.sh <mangled-name substring> "<extra flags>"
#   e.g.  bash scripts/xc_stats.sh xc_fft12_kernelILi9E        bash scripts/xc_stats.sh xc_fft16_kernelILi13E "-DDTX_XC16=1 -DDTX_XC16_ONLY=13"
cd /root/repo/detex_amd/csrc && mkdir -p /tmp/xcasm && \
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize $2 -c dtx_xcfft.hip -o /tmp/xcasm/x.o --save-temps=obj 2>&1 | grep -E "error"
mv /root/repo/detex_amd/csrc/dtx_xcfft-hip* /root/repo/detex_amd/csrc/dtx_xcfft-host* /root/repo/detex_amd/csrc/dtx_xcfft.hip-hip* /tmp/xcasm/ 2>/dev/null
KPAT="$1" python3 - <<'PY'
import os, re
pat=os.environ['KPAT']
s=open('/tmp/xcasm/dtx_xcfft-hip-amdgcn-amd-amdhsa-gfx950.s').read()
for m in re.finditer(r'\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)', s, re.S):
    if pat in m.group(1): print(m.group(1)[-44:], 'scratch', m.group(2), 'sgpr', m.group(3), 'vgpr', m.group(4), 'spill', m.group(5))
name=[l for l in re.findall(r'^(_ZN\S*):', s, re.M) if pat in l][0]
i=s.index(name+':'); j=s.index('.Lfunc_end',i)
blocks=[]; cur=['entry',[]]; blocks.append(cur)
for l in s[i:j].split('\n'):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: cur=[m.group(1),[]]; blocks.append(cur)
    else: cur[1].append(l.strip())
for nm,ins in blocks:
    c=lambda p: sum(1 for x in ins if x.startswith(p))
    n=len([x for x in ins if x and not x.startswith(';') and not x.startswith('.')])
    if n>100: print(nm,'n',n,'scr_st',c('scratch_store'),'scr_ld',c('scratch_load'),'v_fma',c('v_fma'),'pk',c('v_pk_'),'ds_read',c('ds_read'),'ds_write',c('ds_write'),'gload',c('global_load'),'s_load',c('s_load'),'waitcnt',c('s_waitcnt'),'barrier',c('s_barrier'),'v_mov',c('v_mov'))
PY
