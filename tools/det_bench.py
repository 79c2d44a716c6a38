#!/usr/bin/env python
"""Cost of deterministic mode (ops.set_deterministic) on the U-Net step at the benchmark shape:
    python tools/det_bench.py [--size 160] [--steps 10] [--dtype f32|bf16]
prints ms per forward + backward + Adam step with the switch off and on, and checks that two deterministic runs agree bitwise."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from synthsr_amd import ops  # noqa: E402
from synthsr_amd.unet import unet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=160)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--dtype', default='f32')
    a = ap.parse_args()
    S = a.size
    g = torch.Generator().manual_seed(0)
    x = torch.rand(S, S, S, 2, generator=g).cuda()
    t = torch.rand(S ** 3, generator=g).cuda()
    res = {}
    for det in (False, True, True):
        ops.set_deterministic(det)
        net = unet(24, [S, S, S, 2], 5, 3, 1, feat_mult=2, nb_conv_per_level=2, final_pred_activation='linear', batch_norm=-1,
                   seed=0, dtype=a.dtype)

        def step():
            net.loss_l1(x, t)
            net.backward()
            net.adam_step(1e-4)
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        key = 'det' if det else 'default'
        if det and 'det' in res:
            same = torch.equal(res['det'][1], net.params)
            print('deterministic rerun: %.2f ms per step, weights bit-identical to the first deterministic run: %s' % (ms, same))
            assert same
            res['det_warm'] = ms   # the first run also grows the library's private gradient planes (hipMalloc + sync)
        else:
            res[key] = (ms, net.params.clone())
            print('%s %d^3 %s: %.2f ms per U-Net step (status %d)' % (key, S, a.dtype, ms, ops.deterministic_status()))
        del net
    ops.set_deterministic(False)
    print('deterministic / default = %.2fx (steady state: the rerun)' % (res.get('det_warm', res['det'][0]) / res['default'][0]))


if __name__ == '__main__':
    main()
