#!/usr/bin/env python
"""Micro-benchmark of the conv kernels on the U-Net's layer shapes (GPU): python tools/conv_bench.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthsr_amd import ops

LAYERS = [(160, 2, 24), (160, 24, 24), (160, 72, 24), (80, 24, 48), (80, 48, 48), (80, 144, 48), (40, 48, 96), (40, 96, 96),
          (40, 288, 96), (20, 96, 192), (20, 192, 192), (20, 576, 192), (10, 192, 384), (10, 384, 384)]


def t(fn, reps):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    only = sys.argv[2] if len(sys.argv) > 2 else ''
    tot = {'fwd': 0, 'dgrad': 0, 'wgrad': 0}
    print('%-18s %10s %8s %10s %8s %10s %8s' % ('layer', 'fwd ms', 'TF', 'dgrad ms', 'TF', 'wgrad ms', 'TF'))
    for D, ci, co in LAYERS:
        if only and only != '%d_%d_%d' % (D, ci, co):
            continue
        x = torch.randn(D, D, D, ci, device='cuda')
        w = torch.randn(3, 3, 3, ci, co, device='cuda') * 0.05
        b = torch.randn(co, device='cuda')
        dy = torch.randn(D, D, D, co, device='cuda')
        y = torch.empty(D, D, D, co, device='cuda')
        dx = torch.empty(D, D, D, ci, device='cuda')
        dw = torch.zeros_like(w)
        wp, wpd = ops.pack_conv_weights(w, (D, D, D), 0), ops.pack_conv_weights(w, (D, D, D), 1)
        fl = 2.0 * 27 * ci * co * D ** 3
        tf = t(lambda: ops.conv3d(x, wp, b, co, 1, out=y), reps)
        td = t(lambda: ops.conv3d(dy, wpd, None, ci, 0, out=dx), reps) if ci > 2 else float('nan')
        tw = t(lambda: ops.conv3d_wgrad(x, dy, dw), reps)
        tot['fwd'] += tf; tot['wgrad'] += tw; tot['dgrad'] += (td if ci > 2 else 0)
        print('%4d^3 %4d->%-4d %10.3f %8.1f %10.3f %8.1f %10.3f %8.1f' % (D, ci, co, tf, fl / tf / 1e9, td, fl / td / 1e9, tw,
                                                                          fl / tw / 1e9))
    print('sum ms (one call per distinct layer shape):', {k: round(v, 2) for k, v in tot.items()})


if __name__ == '__main__':
    main()
