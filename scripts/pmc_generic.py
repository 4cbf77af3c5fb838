"""Per-kernel HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same program), for
kernels whose demangled name matches a regular expression:
    python scripts/pmc_generic.py <fetch counter_collection.csv> <write counter_collection.csv> '<regex>' [encounters per launch]  > out.json
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB and on gfx950 FETCH_SIZE reports half of a wide coalesced read
stream (MI355X_MICROARCH.md, HBM section) -- the same accounting as scripts/pmc_traffic.py."""
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bench import csrc_sha16          # noqa: E402  (the kernel sources the counters were taken on)


def short(name):
    m = re.search(r'dic::(\w+)', name)
    return m.group(1) if m else name[:60]


def per_kernel(path, counter, pat):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter or not pat.search(r['Kernel_Name']):
            continue
        a = acc.setdefault(short(r['Kernel_Name']), [0, 0.0])
        a[0] += 1
        a[1] += float(r['Counter_Value'])
    return {k: (v[0], v[1] / v[0]) for k, v in acc.items()}


pat = re.compile(sys.argv[3])
fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE', pat), per_kernel(sys.argv[2], 'WRITE_SIZE', pat)
out = {'_note': 'HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024, averaged over the launches of each kernel (rocprofv3 --pmc, separate passes)',
       '_csrc_sha16': csrc_sha16()}
if len(sys.argv) > 4:
    out['_batch'] = int(sys.argv[4])          # encounters per launch of the profiled run (every launch the same: no short last batch)
for k in sorted(set(fetch) & set(write)):
    out[k] = {'launches': fetch[k][0], 'fetch_size_kb': round(fetch[k][1], 1), 'write_size_kb': round(write[k][1], 1),
              'hbm_bytes': int((2 * fetch[k][1] + write[k][1]) * 1024)}
print(json.dumps(out, indent=1))
