#!/usr/bin/env python
"""data gradient of the up-sampled channel range of a folded decoder conv under the library named by SYNTHSR_HIP_LIB: MD5 of the
result (two builds that stage the same values differently must agree bit for bit) and time per launch.   python tools/updgrad_ab.py [reps]"""
import hashlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthsr_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for shape, cl, co, cs in [((80, 80, 80), 48, 24, 24), ((40, 40, 40), 96, 48, 48), ((20, 20, 20), 192, 96, 96), ((42, 38, 50), 48, 24, 24),
                          ((38, 42, 50), 96, 48, 8), ((41, 39, 35), 16, 16, 8), ((24, 24, 32), 96, 48, 48)]:
    g = torch.Generator(device='cpu').manual_seed(sum(shape) + cl)
    hi = tuple(2 * v for v in shape)
    dz = torch.randn(*hi, co, generator=g).cuda()
    w = (torch.randn(3, 3, 3, cs + cl, co, generator=g) * 0.05).cuda()
    wpd = ops.pack_conv_weights_ex(w, shape, cs, cl, 1, up=True)
    split = ops.conv_runs_split('conv3d_up_dgrad', shape, cl, co)
    y = ops.conv3d_up_dgrad(dz, wpd, cl)
    md5 = hashlib.md5(y.cpu().numpy().tobytes()).hexdigest()[:12]
    out = torch.empty(*shape, cl, device='cuda')
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.conv3d_up_dgrad(dz, wpd, cl, out=out)
    e.record()
    torch.cuda.synchronize()
    print('%-14s %3d<-%3d split=%d  md5 %s  %.4f ms' % ('x'.join(map(str, shape)), cl, co, split, md5, s.elapsed_time(e) / reps), flush=True)
