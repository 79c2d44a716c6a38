#!/usr/bin/env python
"""Interleaved A/B timing of forward-conv variants selected with synthsr_conv3d_set_option:
    python tools/ab.py D Cin Cout "4=0" "4=1" ...   (each argument 'opt=val[,opt=val]' is one variant)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthsr_amd import ops, _lib

lib = _lib.load()
DEFAULTS = {0: 1, 1: 0, 2: 0, 3: 0, 4: 1, 5: 1024, 6: 1}
D, ci, co = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
variants = [dict((int(a), int(b)) for a, b in (kv.split('=') for kv in v.split(','))) for v in sys.argv[4:]] or [{}]
x = torch.randn(D, D, D, ci, device='cuda')
w = torch.randn(3, 3, 3, ci, co, device='cuda') * .05
b = torch.randn(co, device='cuda')
y = torch.empty(D, D, D, co, device='cuda')
fl = 2.0 * 27 * ci * co * D ** 3


def setopts(o):
    for k, v in DEFAULTS.items():
        lib.synthsr_conv3d_set_option(k, o.get(k, v))


packed = []
for o in variants:
    setopts(o)
    packed.append(ops.pack_conv_weights(w, (D, D, D), 0))
best = [1e9] * len(variants)
for rnd in range(6):
    for i, o in enumerate(variants):
        setopts(o)
        ops.conv3d(x, packed[i], b, co, 1, out=y)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            ops.conv3d(x, packed[i], b, co, 1, out=y)
        e.record()
        torch.cuda.synchronize()
        best[i] = min(best[i], s.elapsed_time(e) / 10)
for o, ms in zip(variants, best):
    print('%d^3 %d->%d %-14s %.4f ms  %.1f TF' % (D, ci, co, o, ms, fl / ms / 1e9))
setopts({})
