#!/usr/bin/env python
"""Copy the judged summaries of gpurun_out/prof_<tag>/ into profiles/:
   python scripts/summarize_profile.py <tag> <kernel-substring>"""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

tag, kern = sys.argv[1], sys.argv[2]
src = 'gpurun_out/prof_%s' % tag
shutil.copy(glob.glob(src + '/trace/runc/*_kernel_stats.csv')[0], 'profiles/%s_bench_kernel_stats.csv' % tag)
out = {}
full_names = collections.Counter()
for d in ['pmc_fetch', 'pmc_write', 'pmc_sq', 'pmc_sq2']:
    fs = glob.glob('%s/%s/runc/*counter_collection.csv' % (src, d))
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if kern in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            full_names[r['Kernel_Name']] += 1
    for k, v in agg.items():
        out[k] = {'launches': len(v), 'mean_per_launch': sum(v) / len(v)}
if 'FETCH_SIZE' in out and 'WRITE_SIZE' in out:
    # MI355X_MICROARCH.md HBM section: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE
    # reports half the bytes of wide coalesced reads -> doubled
    out['hbm_bytes_per_launch_corrected'] = (2 * out['FETCH_SIZE']['mean_per_launch'] +
                                             out['WRITE_SIZE']['mean_per_launch']) * 1024


def source_digest():
    """sha1 over the kernel sources: bench.py compares it with the tree it runs from, so that a committed
    counter summary of an OLDER kernel is flagged instead of silently reported."""
    h = hashlib.sha1()
    for fn in sorted(glob.glob('detex_amd/csrc/*.hip') + glob.glob('detex_amd/csrc/*.h')):
        h.update(os.path.basename(fn).encode())
        h.update(open(fn, 'rb').read())
    return h.hexdigest()


json.dump({'kernel': kern, 'tag': tag, 'kernel_instances': dict(full_names), 'source_sha1': source_digest(),
           'counters': out},
          open('profiles/%s_bench_pmc_%s.json' % (tag, kern), 'w'), indent=1)
print(json.dumps(out, indent=1))
