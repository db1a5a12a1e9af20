#!/usr/bin/env python
"""Copy the judged summaries of gpurun_out/prof_<tag>/ (scripts/profile_xc.sh) into profiles/:
   python scripts/summarize_xc.py <tag>
Writes profiles/<tag>_xcorr_kernel_stats.csv and profiles/<tag>_xcorr_pmc.json (mean per launch of
every counter for the lag-search kernel (xc_fft22_kernel / xc_fft12_kernel) and xc_finish_kernel; FETCH_SIZE doubled per the gfx950 note of
MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = 'gpurun_out/prof_%s' % tag
st = glob.glob(src + '/trace/**/*_kernel_stats.csv', recursive=True)
if st:
    shutil.copy(st[0], 'profiles/%s_xcorr_kernel_stats.csv' % tag)
res = {}
for kern in ("xc_fft22_kernel", "xc_fft_kernel", "xc_finish"):
    out = {}
    for d in ['pmc_fetch', 'pmc_write', 'pmc_tcc', 'pmc_sq', 'pmc_sq2']:
        for f in glob.glob('%s/%s/**/*counter_collection.csv' % (src, d), recursive=True):
            agg = collections.defaultdict(lambda: collections.defaultdict(float))
            for r in csv.DictReader(open(f)):
                if kern in r['Kernel_Name']:
                    agg[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
            for k, v in agg.items():
                # the first launch of the process is the warm-up (workspaces are allocated): keep the later ones
                vals = [v[i] for i in sorted(v, key=int)]
                vals = vals[1:] if len(vals) > 1 else vals
                out[k] = {'launches': len(vals), 'mean_per_launch': sum(vals) / len(vals)}
    if 'FETCH_SIZE' in out and 'WRITE_SIZE' in out:
        out['hbm_bytes_per_launch_corrected'] = (2 * out['FETCH_SIZE']['mean_per_launch'] +
                                                 out['WRITE_SIZE']['mean_per_launch']) * 1024
    if out:
        res[kern] = out
h = hashlib.sha1()          # the kernel sources the counters belong to (bench.py flags a stale summary)
for fn in sorted(glob.glob('detex_amd/csrc/*.hip') + glob.glob('detex_amd/csrc/*.h')):
    h.update(os.path.basename(fn).encode())
    h.update(open(fn, 'rb').read())
json.dump({'tag': tag, 'source_sha1': h.hexdigest(), 'kernels': res}, open('profiles/%s_xcorr_pmc.json' % tag, 'w'), indent=1)
print(json.dumps(res, indent=1))
