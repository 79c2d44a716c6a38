#!/usr/bin/env python
"""weight-gradient kernels of the benchmark network with and without their final atomic flush (debug option 1, bit 8):
how much of each launch is the cross-workgroup accumulation.  python tools/wgrad_flush_ab.py [size]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from synthsr_amd import ops, _lib  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 160
LAYERS = [(S, 24, 24), (S // 2, 24, 48), (S // 2, 48, 48), (S // 4, 48, 96), (S // 4, 96, 96), (S // 8, 96, 192),
          (S // 8, 192, 192), (S // 16, 192, 384), (S // 16, 384, 384), (S // 8, 576, 192), (S // 4, 288, 96), (S // 2, 144, 48)]
lib = _lib.load()


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for D, ci, co in LAYERS:
    x = torch.randn(D, D, D, ci, device='cuda')
    dz = torch.randn(D, D, D, co, device='cuda')
    dw = torch.zeros(3, 3, 3, ci, co, device='cuda')
    db = torch.zeros(co, device='cuda')
    res = []
    for dbg in (0, 8):
        lib.synthsr_conv3d_set_option(1, dbg)
        res.append(timeit(lambda: ops.conv3d_wgrad(x, dz, dw, db)))
    lib.synthsr_conv3d_set_option(1, 0)
    gf = 2 * 27 * ci * co * D ** 3 / 1e9
    print('%3d^3 %3d->%3d  wgrad %7.1f us (%5.1f TF)   without flush %7.1f us   flush share %4.1f %%' % (
        D, ci, co, res[0], gf / (res[0] * 1e-6) / 1e3, res[1], 100 * (res[0] - res[1]) / res[0]))
