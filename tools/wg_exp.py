#!/usr/bin/env python
"""weight-gradient timing experiments: python tools/wg_exp.py D Cin Cout [dbg values...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthsr_amd import ops, _lib
from conv_bench import t
lib = _lib.load()
D, ci, co = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(D, D, D, ci, device='cuda'); dy = torch.randn(D, D, D, co, device='cuda')
dw = torch.zeros(3, 3, 3, ci, co, device='cuda')
fl = 2.0 * 27 * ci * co * D ** 3
for dbg in [int(v) for v in sys.argv[4:]] or [0]:
    lib.synthsr_conv3d_set_option(1, dbg)
    ms = min(t(lambda: ops.conv3d_wgrad(x, dy, dw), 10) for _ in range(4))
    print('%d^3 %d->%d dbg=%d  %.4f ms  %.1f TF' % (D, ci, co, dbg, ms, fl / ms / 1e9))
lib.synthsr_conv3d_set_option(1, 0)
if os.environ.get('WG_TOTALS'):
    for tot in [int(v) for v in os.environ['WG_TOTALS'].split(',')]:
        lib.synthsr_conv3d_set_option(2, tot)
        ms = min(t(lambda: ops.conv3d_wgrad(x, dy, dw), 10) for _ in range(4))
        print('%d^3 %d->%d total=%d  %.4f ms  %.1f TF' % (D, ci, co, tot, ms, fl / ms / 1e9))
    lib.synthsr_conv3d_set_option(2, 0)
