#!/usr/bin/env python
"""Registers / spills / occupancy of every kernel of one HIP source, from hipcc -Rpass-analysis=kernel-resource-usage:

    python tools/kernel_resources.py synthsr_amd/csrc/conv_split.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ['hipcc', '-x', 'hip', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', '/dev/null',
       '-Rpass-analysis=kernel-resource-usage'] + sys.argv[2:]
err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r'Function Name: (\S+)', line) or re.search(r' Name: (\S+)', line)
    if m:
        cur = {'name': m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)', line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
try:
    names = subprocess.run(['c++filt'] + [r['name'] for r in rows], stdout=subprocess.PIPE, text=True).stdout.splitlines()
except OSError:
    names = [r['name'] for r in rows]
print('%-70s %5s %5s %6s %6s %5s %4s' % ('kernel', 'VGPR', 'AGPR', 'vspill', 'sspill', 'scr', 'occ'))
for r, n in zip(rows, names):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'\(.*', '', n).replace('void ', '')
    print('%-70s %5s %5s %6s %6s %5s %4s' % (n[:70], r.get('VGPRs'), r.get('AGPRs'), r.get('VGPRs Spill'), r.get('SGPRs Spill'),
                                             r.get('ScratchSize'), r.get('Occupancy')))
