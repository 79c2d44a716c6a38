#!/usr/bin/env python
"""timing of the reduction-type pointwise kernels at level-0 size: python tools/pw_exp.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthsr_amd import ops
from conv_bench import t
D, C = 160, 24
x = torch.randn(D, D, D, C, device='cuda'); g = torch.randn(D, D, D, C, device='cuda')
stats = torch.zeros(2 * C, device='cuda'); ws = torch.zeros(2 * C, dtype=torch.float64, device='cuda')
sums = torch.zeros(2 * C, device='cuda'); gamma = torch.ones(C, device='cuda'); db = torch.zeros(C, device='cuda')
out = torch.empty_like(x)
ops.bn_stats(x, stats, ws)
gb = x.numel() * 4 / 1e9
for name, fn, passes in [('bn_stats', lambda: ops.bn_stats(x, stats, ws), 1),
                         ('bn_reduce_bwd', lambda: ops.bn_reduce_bwd(g, x, stats, sums), 2),
                         ('bn_elu_bwd', lambda: ops.bn_elu_bwd(g, x, stats, gamma, sums, dbias=db, out=out), 3),
                         ('elu_bwd', lambda: ops.elu_bwd(g, x, dbias=db, out=out), 3)]:
    ms = min(t(fn, 10) for _ in range(3))
    print('%-14s %.4f ms  %.2f TB/s' % (name, ms, passes * gb / ms))
