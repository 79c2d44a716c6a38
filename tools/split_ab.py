#!/usr/bin/env python
"""A/B of the split-arithmetic conv kernels (variant code: units digit = synthsr_conv3d_set_option(8, .): 0 = round-3 kernels,
1 = round-4 kernels, 2 = LDS-weights kernel everywhere; tens digit 1 = stacked 24-channel layout OFF (option 10); >= 100:
option 9 = --min-wgs): per layer shape, time of forward (ELU), forward + BatchNorm statistics, data gradient (x ELU'), weight gradient, and
the largest difference between the two variants' results (same arithmetic: expected ~1 ulp of the accumulation).

    python tools/split_ab.py [--reps 20] [--only 160_24_24,...] [--variants 0,1]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from synthsr_amd import _lib, ops  # noqa: E402


def timeit(fn, reps):
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
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--only', default='')
    ap.add_argument('--variants', default='0,1')
    ap.add_argument('--min-wgs', type=int, default=0, help='option 9 per variant: v >= 100 means variant v - 100 with this threshold')
    ap.add_argument('--no-wgrad', action='store_true')
    a = ap.parse_args()
    lib = _lib.load()
    variants = [int(v) for v in a.variants.split(',')]
    torch.manual_seed(0)
    print('%-16s %-8s' % ('layer', 'kernel') + ' '.join('v%d ms ' % v for v in variants) + '  max |diff| / rms')
    for D, ci, co in [(160, 24, 24), (80, 24, 48), (80, 48, 48), (80, 48, 24), (40, 48, 96), (40, 96, 96), (40, 96, 48),
                      (20, 96, 192), (20, 192, 192), (20, 192, 96), (10, 192, 384), (10, 384, 384)]:
        if a.only and '%d_%d_%d' % (D, ci, co) not in a.only.split(','):
            continue
        shape = (D, D, D)
        x = torch.randn(D, D, D, ci, device='cuda')
        w = torch.randn(3, 3, 3, ci, co, device='cuda') * 0.05
        b = torch.randn(co, device='cuda')
        dy = torch.randn(D, D, D, co, device='cuda')
        res, tm = {}, {}
        for v in variants:
            lib.synthsr_conv3d_set_option(8, v % 10)
            lib.synthsr_conv3d_set_option(10, 0 if (v // 10) % 10 == 1 else 1)   # tens digit 1: stacked 24-channel layout off
            lib.synthsr_conv3d_set_option(9, a.min_wgs if v >= 100 and a.min_wgs else 0)
            wp, wpd = ops.pack_conv_weights(w, shape, 0), ops.pack_conv_weights(w, shape, 1)
            y = torch.empty(D, D, D, co, device='cuda')
            ys = torch.empty(D, D, D, co, device='cuda')
            dx = torch.empty(D, D, D, ci, device='cuda')
            dw = torch.zeros_like(w)
            stats = torch.zeros(2 * co, device='cuda')
            ws = torch.zeros(512 * 2 * co + 1024, device='cuda')
            f_fwd = lambda: ops.conv3d(x, wp, b, co, 1, out=y)
            f_st = lambda: ops.conv3d_stats(x, wp, b, co, stats, ws, out=ys)
            f_dg = (lambda: ops.conv3d_add(dy, wpd, None, x, ci, 2, out=dx)) if ci == co else \
                   (lambda: ops.conv3d(dy, wpd, None, ci, 0, out=dx))
            f_wg = lambda: ops.conv3d_wgrad(x, dy, dw)
            tm[v] = (timeit(f_fwd, a.reps), timeit(f_st, a.reps), timeit(f_dg, a.reps), timeit(f_wg, a.reps))
            dw.zero_()
            f_wg()
            res[v] = (y.clone(), ys.clone(), dx.clone(), dw.clone(), stats.clone())
        lib.synthsr_conv3d_set_option(8, 1)
        lib.synthsr_conv3d_set_option(10, 1)
        lib.synthsr_conv3d_set_option(9, 0)
        for k, nm in enumerate(('fwd', 'fwd+st', 'dgrad', 'wgrad')):
            d = ''
            if len(variants) > 1:
                r0, r1 = res[variants[0]][k], res[variants[-1]][k]
                d = '%.2e' % (float((r0 - r1).abs().max()) / float(r0.pow(2).mean().sqrt()))
                if k == 1:
                    s0, s1 = res[variants[0]][4], res[variants[-1]][4]
                    d += '  stats %.2e' % (float((s0 - s1).abs().max()) / float(s0.abs().max()))
            print('%4d^3 %3d->%-3d %-8s' % (D, ci, co, nm) + ' '.join('%6.3f' % tm[v][k] for v in variants) + '   ' + d)


if __name__ == '__main__':
    main()
