#!/usr/bin/env python
"""Per-kernel average of one rocprofv3 --pmc counter (csv output):
    python tools/pmc_summary.py gpurun_out/pmc_f/f_counter_collection.csv [min_avg]"""
import csv, re, sys, collections
rows = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        name = re.sub(r'^void ', '', name)
        name = re.sub(r'\(.*', '', name)
        rows[(name, r['Grid_Size'], r['Counter_Name'])].append(float(r['Counter_Value']))
out = []
for (name, grid, cn), v in rows.items():
    out.append((sum(v), name, grid, cn, len(v), sum(v) / len(v), max(v)))
print('%-60s %10s %12s %6s %14s %14s' % ('kernel', 'grid', 'counter', 'calls', 'avg', 'max'))
for tot, name, grid, cn, n, avg, mx in sorted(out, reverse=True)[:40]:
    print('%-60s %10s %12s %6d %14.1f %14.1f' % (name[:60], grid, cn, n, avg, mx))
