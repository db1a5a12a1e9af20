#!/usr/bin/env python
"""Copy the judged summaries of gpurun_out/prof_<tag>/ into profiles/:
   python scripts/summarize_profile.py <tag> <kernel-substring>"""
import collections
import csv
import glob
import json
import shutil
import sys

tag, kern = sys.argv[1], sys.argv[2]
src = 'gpurun_out/prof_%s' % tag
shutil.copy(glob.glob(src + '/trace/runc/*_kernel_stats.csv')[0], 'profiles/%s_bench_kernel_stats.csv' % tag)
out = {}
for d in ['pmc_fetch', 'pmc_write', 'pmc_sq', 'pmc_sq2']:
    fs = glob.glob('%s/%s/runc/*counter_collection.csv' % (src, d))
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if kern in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        out[k] = {'launches': len(v), 'mean_per_launch': sum(v) / len(v)}
if 'FETCH_SIZE' in out and 'WRITE_SIZE' in out:
    # MI355X_MICROARCH.md HBM section: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE
    # reports half the bytes of wide coalesced reads -> doubled
    out['hbm_bytes_per_launch_corrected'] = (2 * out['FETCH_SIZE']['mean_per_launch'] +
                                             out['WRITE_SIZE']['mean_per_launch']) * 1024
json.dump({'kernel': kern, 'tag': tag, 'counters': out},
          open('profiles/%s_bench_pmc_%s.json' % (tag, kern), 'w'), indent=1)
print(json.dumps(out, indent=1))
