#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel stats table (like --stats):
    python tools/rocprof_summary.py gpurun_out/prof1/r1_results.db > profiles/r01_kernel_stats.txt"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name[:110]


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) '
                     'from kernels group by name order by sum(duration) desc').fetchall()
    total = sum(r[2] for r in rows)
    print('# rocprofv3 --kernel-trace summary of %s' % path)
    print('# total kernel time %.3f ms over %d dispatches' % (total / 1e6, sum(r[1] for r in rows)))
    print('%-112s %8s %12s %12s %12s %12s %7s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', 'pct'))
    for n, cnt, tot, avg, mn, mx in rows:
        print('%-112s %8d %12.3f %12.2f %12.2f %12.2f %6.2f%%' % (short(n), cnt, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3,
                                                                 100.0 * tot / total))


if __name__ == '__main__':
    main(sys.argv[1])
