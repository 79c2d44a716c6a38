#!/usr/bin/env python
"""Weight gradient of the up-sampled channel range of the four folded decoder convs of the benchmark network: time per launch
(torch.cuda.Event over `reps` launches) under both arithmetics, and the split result's distance from the fp32-MFMA one.

    python tools/upwgrad_bench.py [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthsr_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for D, cl, co in [(80, 48, 24), (40, 96, 48), (20, 192, 96), (10, 384, 192)]:
    g = torch.Generator(device='cpu').manual_seed(D)
    lo = torch.randn(D, D, D, cl, generator=g).cuda()
    dz = torch.randn(2 * D, 2 * D, 2 * D, co, generator=g).cuda()
    out = {}
    for mode in ('fp32_mfma', 'split'):
        ops.set_conv_arithmetic(mode)
        dw = torch.zeros(3, 3, 3, cl, co, device='cuda')
        dwc = torch.empty(8, 27, cl, co, device='cuda')
        ops.conv3d_up_wgrad(lo, dz, dwc, dw, 0)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            ops.conv3d_up_wgrad(lo, dz, dwc, dw, 0)
        e.record()
        torch.cuda.synchronize()
        dw.zero_()
        ops.conv3d_up_wgrad(lo, dz, dwc, dw, 0)
        out[mode] = (s.elapsed_time(e) / reps, dw.clone())
    a, b = out['fp32_mfma'][1], out['split'][1]
    gf = 2 * 64 * cl * co * D ** 3 / 1e9
    print('%3d^3 %3d->%3d  fp32_mfma %.3f ms (%.0f TF)  split %.3f ms (%.0f TF = %.2f of 417)  max|diff| %.2e of rms %.3e'
          % (D, cl, co, out['fp32_mfma'][0], gf / out['fp32_mfma'][0], out['split'][0], gf / out['split'][0],
             gf / out['split'][0] / 416.7, float((a - b).abs().max()), float(a.pow(2).mean().sqrt())), flush=True)
ops.set_conv_arithmetic('split')
