#!/usr/bin/env python
"""A/B of the split weight-gradient kernels in one process on one GPU (option 12 of include/synthsr_hip_tuning.h):

    python tools/wgrad_ab.py stack      # bit 0: stacked column tiles of the 24-column kernel        (option 12 = 0 | 1)
    python tools/wgrad_ab.py col24      # bit 1: 24-column workgroups also where 48 divides Cout     (1 | 3)
    python tools/wgrad_ab.py ciw16      # bit 2: 8 | 16 input channels per 48-column workgroup       (5 | 1)
    python tools/wgrad_ab.py ciw24      # bit 3: 8 | all 24 input channels per 24-column workgroup   (9 | 1)
    (round 5: `var24` compared layout / schedule variants of the all-channels kernel behind option 12 bits 4-6 at commit 9f-series;
     the winner -- channel-quad planes + 42 row-tile slots -- is now THE kernel, the record is profiles/r05_split_wgrad_var24_ab.txt)

For each setting: the error of dW / dbias against a float64 convolution gradient (torch, small volumes with ragged edges)
relative to the largest |dW| / |dbias|, then the time per launch (torch.cuda.Event over 20 launches, with the bias gradient)
of the layer shapes the setting touches, two passes.  The records under profiles/r04_split_wgrad_*_ab.txt are this output.
Inside the training step: SYNTHSR_CONV_OPTIONS=12=<value> python bench.py --steps 40 --warmup 5 --no-cpu-baseline
--no-arith-compare."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from synthsr_amd import _lib, ops

CASES = {   # name -> (option values (off, on), accuracy shapes [(D, ci, co)], timing shapes [(D, ci, co)])
    'stack': ((0, 1), [((16, 20, 32), 24, 24)], [(160, 24, 24), (80, 24, 48)]),
    'col24': ((1, 3), [((8, 12, 32), 48, 48)], [(80, 48, 48), (80, 24, 48), (40, 96, 96), (40, 48, 96), (20, 192, 192), (10, 384, 384)]),
    'ciw16': ((5, 1), [((8, 12, 32), 48, 48), ((5, 7, 19), 32, 96), ((4, 4, 16), 16, 48)],
              [(80, 48, 48), (40, 96, 96), (40, 48, 96), (20, 192, 192), (20, 96, 192), (10, 384, 384)]),
    'ciw24': ((9, 1), [((8, 12, 32), 24, 24), ((5, 7, 19), 24, 24), ((4, 4, 16), 24, 24)], [(160, 24, 24)]),
    'var24': ((1, 1 + 16, 1 + 48, 1 + 80, 1 + 112), [((8, 12, 32), 24, 24), ((5, 7, 19), 24, 24), ((4, 4, 16), 24, 24), ((20, 20, 20), 24, 24)],
              [(160, 24, 24)]),
}


def timed(fn, reps=20):
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
    which = sys.argv[1] if len(sys.argv) > 1 else 'ciw24'
    opts, acc_shapes, time_shapes = CASES[which]
    lib = _lib.load()
    torch.manual_seed(0)
    for D, ci, co in acc_shapes:
        x = torch.randn(*D, ci, device='cuda')
        dy = torch.randn(*D, co, device='cuda')
        w = torch.zeros(co, ci, 3, 3, 3, dtype=torch.float64, device='cuda', requires_grad=True)
        F.conv3d(x.double().permute(3, 0, 1, 2)[None], w, padding=1).backward(dy.double().permute(3, 0, 1, 2)[None])
        ref = w.grad.permute(2, 3, 4, 1, 0)
        rb = dy.double().sum((0, 1, 2))
        first = None
        for o in opts:
            lib.synthsr_conv3d_set_option(12, o)
            dw = torch.zeros(3, 3, 3, ci, co, device='cuda')
            db = torch.zeros(co, device='cuda')
            ops.conv3d_wgrad(x, dy, dw, db)
            line = '%s %d->%d option 12 = %d: dW err vs float64 %.2e, dbias err %.2e' % (
                D, ci, co, o, float((dw.double() - ref).abs().max() / ref.abs().max()),
                float((db.double() - rb).abs().max() / rb.abs().max()))
            if which == 'var24':   # ordered sums: the same products in the same order -> the same bits
                prev = ops.set_deterministic(True)
                dwd, dbd = torch.zeros_like(dw), torch.zeros_like(db)
                ops.conv3d_wgrad(x, dy, dwd, dbd)
                ops.set_deterministic(prev)
                if first is None:
                    first = (dwd, dbd)
                line += '; deterministic run bit-identical to option %d: %s' % (opts[0], torch.equal(dwd, first[0]) and torch.equal(dbd, first[1]))
            print(line)
    for _ in range(2):
        for o in opts:
            lib.synthsr_conv3d_set_option(12, o)
            res = []
            for D, ci, co in time_shapes:
                x = torch.randn(D, D, D, ci, device='cuda')
                dy = torch.randn(D, D, D, co, device='cuda')
                dw = torch.zeros(3, 3, 3, ci, co, device='cuda')
                db = torch.zeros(co, device='cuda')
                res.append('%d^3 %d->%d %.4f' % (D, ci, co, timed(lambda: ops.conv3d_wgrad(x, dy, dw, db))))
                del x, dy
            print('option 12 = %d: ' % o + ' | '.join(res) + '  (ms)')
    if which == 'var24':   # the full-size layer, ordered sums: every variant against the first one, bit for bit
        D, ci, co = time_shapes[0]
        x = torch.randn(D, D, D, ci, device='cuda')
        dy = torch.randn(D, D, D, co, device='cuda')
        prev = ops.set_deterministic(True)
        first = None
        for o in opts:
            lib.synthsr_conv3d_set_option(12, o)
            dw, db = torch.zeros(3, 3, 3, ci, co, device='cuda'), torch.zeros(co, device='cuda')
            ops.conv3d_wgrad(x, dy, dw, db)
            first = first or (dw, db)
            print('%d^3 option 12 = %d deterministic: bit-identical to option %d: %s (status %d)' % (
                D, o, opts[0], torch.equal(dw, first[0]) and torch.equal(db, first[1]), ops.deterministic_status()))
        ops.set_deterministic(prev)
    lib.synthsr_conv3d_set_option(12, 1)


if __name__ == '__main__':
    main()
