#!/usr/bin/env python
"""forward of the up-sampled channel range of a folded decoder conv under the library named by SYNTHSR_HIP_LIB: MD5 of the results
(two builds that only re-partition the work must agree bit for bit) and time per launch.   python tools/upfwd_ab.py [reps]"""
import hashlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthsr_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for shape, cl, co, cs in [((80, 80, 80), 48, 24, 24), ((40, 40, 40), 96, 48, 48), ((42, 38, 50), 48, 24, 24), ((38, 42, 50), 96, 48, 8), ((41, 39, 35), 16, 16, 8),
                          ((24, 24, 32), 96, 48, 48)]:
    g = torch.Generator(device='cpu').manual_seed(sum(shape) + cl)
    lo = torch.randn(*shape, cl, generator=g).cuda()
    hi = tuple(2 * v for v in shape)
    w = (torch.randn(3, 3, 3, cs + cl, co, generator=g) * 0.05).cuda()
    b = torch.randn(co, generator=g).cuda()
    add = torch.randn(*hi, co, generator=g).cuda()
    wp = ops.pack_conv_weights_ex(w, shape, cs, cl, 0, up=True)
    split = ops.conv_runs_split('conv3d_up_fwd', shape, cl, co)
    outs = []
    for bias, addend, act in ((None, None, 0), (b, add, 1)):
        y = ops.conv3d_up(lo, wp, bias, addend, co, act)
        outs.append(hashlib.md5(y.cpu().numpy().tobytes()).hexdigest()[:12])
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = torch.empty(*hi, co, device='cuda')
    ops.conv3d_up(lo, wp, b, add, co, 1, out=out)
    s.record()
    for _ in range(reps):
        ops.conv3d_up(lo, wp, b, add, co, 1, out=out)
    e.record()
    torch.cuda.synchronize()
    print('%-14s %3d->%3d split=%d  md5 %s %s  %.4f ms' % ('x'.join(map(str, shape)), cl, co, split, outs[0], outs[1], s.elapsed_time(e) / reps), flush=True)
