#!/usr/bin/env python
"""times the bf16 conv kernels on the layer shapes of the benchmark network:  python tools/conv_bf16_bench.py [size]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from synthsr_amd import ops  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 160
LAYERS = [(S, 8, 24), (S, 24, 24), (S, 72, 24), (S // 2, 24, 48), (S // 2, 48, 48), (S // 2, 144, 48), (S // 4, 96, 96),
          (S // 4, 288, 96), (S // 8, 192, 192), (S // 8, 576, 192), (S // 16, 384, 384)]


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for D, ci, co in LAYERS:
    x = torch.randn(D, D, D, ci, device='cuda').bfloat16()
    dz = torch.randn(D, D, D, co, device='cuda').bfloat16()
    w = torch.randn(3, 3, 3, ci, co, device='cuda') * .05
    b = torch.zeros(co, device='cuda')
    wp, wpd = ops.pack_conv_weights_bf16(w, 0), (ops.pack_conv_weights_bf16(w, 1) if co % 8 == 0 else None)
    out = torch.empty(D, D, D, co, device='cuda', dtype=torch.bfloat16)
    dx = torch.empty(D, D, D, ci, device='cuda', dtype=torch.bfloat16)
    dw = torch.zeros(3, 3, 3, ci, co, device='cuda')
    stats = torch.zeros(2 * co, device='cuda')
    gf = 2 * 27 * ci * co * D ** 3 / 1e9
    mb = 2 * D ** 3 * (ci + co) / 1e6
    t_f = timeit(lambda: ops.conv3d_bf16(x, wp, b, co, 1, out=out))
    t_s = timeit(lambda: ops.conv3d_bf16(x, wp, b, co, 1, stats=stats, out=out))
    t_d = timeit(lambda: ops.conv3d_bf16(dz, wpd, None, ci, 0, out=dx))
    t_w = timeit(lambda: ops.conv3d_wgrad_bf16(x, dz, dw))
    print('%3d^3 %3d->%3d  %7.1f GF %7.1f MB | fwd %7.1f us %6.0f TF %5.2f TB/s | +stats %7.1f | dgrad %7.1f us %6.0f TF | '
          'wgrad %7.1f us %6.0f TF' % (D, ci, co, gf, mb, t_f * 1e3, gf / t_f, mb / t_f / 1e3, t_s * 1e3, t_d * 1e3, gf / t_d,
                                     t_w * 1e3, gf / t_w))
