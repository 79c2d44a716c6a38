#!/usr/bin/env python
"""Instruction mix of the kernels in a hipcc -S listing: python tools/isa_mix.py conv3d.s [name-filter]"""
import re, sys, collections
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    c = collections.Counter()
    for line in body.split('\n'):
        line = line.strip()
        if not line or line[0] in '.;/' or line.endswith(':'):
            continue
        op = line.split()[0]
        if op.startswith('v_mfma'): c['mfma'] += 1
        elif op.startswith('v_'): c['valu'] += 1; c['valu:' + op] += 1
        elif op.startswith('ds_'): c['ds'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): c['vmem'] += 1
        elif op.startswith('s_waitcnt'): c['waitcnt'] += 1
        elif op.startswith('s_nop'): c['nop'] += 1
        elif op.startswith('s_barrier'): c['barrier'] += 1
        elif op.startswith('s_load'): c['smem'] += 1
        elif op.startswith('s_'): c['salu'] += 1
    short = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', name)[:60]
    print(short, {k: v for k, v in c.items() if ':' not in k})
    print('   ', sorted([(v, k[5:]) for k, v in c.items() if ':' in k], reverse=True)[:12])
