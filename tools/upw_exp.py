#!/usr/bin/env python
"""timing of the parity-conv weight gradient (synthsr_conv3d_up_wgrad): python tools/upw_exp.py [lo Cl Cout]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthsr_amd import ops, _lib
from conv_bench import t
lib = _lib.load()
lo, Cl, Co = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (80, 48, 24)
x = torch.randn(lo, lo, lo, Cl, device='cuda'); dz = torch.randn(2 * lo, 2 * lo, 2 * lo, Co, device='cuda')
dwc = torch.empty(8, 27, Cl, Co, device='cuda'); dw = torch.zeros(3, 3, 3, Cl + 24, Co, device='cuda')
fl = 8.0 * lo ** 3 * 8 * Cl * Co * 2
for p4 in (1, 0, 1, 0):
    lib.synthsr_conv3d_set_option(4, p4)
    ms = min(t(lambda: ops.conv3d_up_wgrad(x, dz, dwc, dw, 24), 10) for _ in range(4))
    print('lo %d^3 %d->%d p4=%d  %.4f ms  %.1f TF (folded flops)' % (lo, Cl, Co, p4, ms, fl / ms / 1e9))
lib.synthsr_conv3d_set_option(4, 1)
