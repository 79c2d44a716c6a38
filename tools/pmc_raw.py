#!/usr/bin/env python
"""per-kernel sums of every counter of a rocprofv3 --pmc csv, as shares of GRBM_GUI_ACTIVE / 8 (cycles of one XCD) x 256 CUs
where that makes sense:   python tools/pmc_raw.py <counter_collection.csv> [substring of the kernel name]"""
import collections
import csv
import re
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*', '', name)
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    key = (name, r['Grid_Size'])
    rows[key][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        calls[key] += 1
        rows[key]['_us'] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
for key, c in sorted(rows.items(), key=lambda kv: -kv[1]['_us']):
    n = max(calls[key], 1)
    cyc = c.get('GRBM_GUI_ACTIVE', 0.0) / 8
    print('%s grid %s calls %d avg %.1f us  clock %.2f GHz' % (key[0][:70], key[1], n, c['_us'] / n, cyc / max(c['_us'], 1e-9) * 1e-3))
    for k, v in sorted(c.items()):
        if k.startswith('_') or k == 'GRBM_GUI_ACTIVE':
            continue
        print('    %-28s %14.0f per call   %.3f per CU-cycle' % (k, v / n, v / max(cyc * 256, 1)))
