#!/usr/bin/env python
"""A/B of forward-conv launch options (timing only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthsr_amd import ops, _lib
from conv_bench import t

lib = _lib.load()
DEFAULTS = {0: 1, 1: 0, 2: 0, 3: 0}
for D, ci, co in [(160, 24, 24), (80, 48, 24), (80, 144, 48)]:
    x = torch.randn(D, D, D, ci, device='cuda'); w = torch.randn(3, 3, 3, ci, co, device='cuda') * .05
    b = torch.randn(co, device='cuda'); y = torch.empty(D, D, D, co, device='cuda')
    fl = 2.0 * 27 * ci * co * D ** 3
    for name, opts in [('default', {}), ('hybrid', {3: 1}), ('no-persist', {0: 0})]:
        for k, v in DEFAULTS.items():
            lib.synthsr_conv3d_set_option(k, v)
        for k, v in opts.items():
            lib.synthsr_conv3d_set_option(k, v)
        wp = ops.pack_conv_weights(w, (D, D, D), 0)
        ms = min(t(lambda: ops.conv3d(x, wp, b, co, 1, out=y), 5) for _ in range(3))
        print('%d^3 %d->%d %-22s %.3f ms  %.1f TF' % (D, ci, co, name, ms, fl / ms / 1e9))
for k, v in DEFAULTS.items():
    lib.synthsr_conv3d_set_option(k, v)
