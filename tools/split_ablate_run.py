#!/usr/bin/env python
"""times the 160^3 24->24 forward (ELU) and data gradient (x ELU') of every ablation build under tools/scratch (one process each)"""
import glob, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == '--one':
    sys.path.insert(0, R)
    import torch
    from synthsr_amd import _lib
    _lib.LIB_PATH = sys.argv[2]
    from synthsr_amd import ops
    D, ci, co = [int(v) for v in sys.argv[3].split('_')]
    x = torch.randn(D, D, D, ci, device='cuda'); w = torch.randn(3, 3, 3, ci, co, device='cuda') * .05
    b = torch.randn(co, device='cuda'); dy = torch.randn(D, D, D, co, device='cuda')
    wp, wpd = ops.pack_conv_weights(w, (D, D, D), 0), ops.pack_conv_weights(w, (D, D, D), 1)
    y = torch.empty(D, D, D, co, device='cuda'); dx = torch.empty(D, D, D, ci, device='cuda')
    def t(fn, reps=20):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / reps
    f = t(lambda: ops.conv3d(x, wp, b, co, 1, out=y))
    g = t(lambda: ops.conv3d_add(dy, wpd, None, x, ci, 2, out=dx)) if ci == co else t(lambda: ops.conv3d(dy, wpd, None, ci, 0, out=dx))
    print('%-40s %s fwd %.3f ms  dgrad %.3f ms' % (os.path.basename(sys.argv[2]), sys.argv[3], f, g), flush=True)
else:
    libs = [os.path.join(R, 'synthsr_amd', 'libsynthsr_hip.so')] + sorted(glob.glob(os.path.join(R, 'tools', 'scratch', 'libsynthsr_abl_*.so')))
    for shape in (sys.argv[1:] or ['160_24_24']):
        for lib in libs:
            subprocess.run([sys.executable, __file__, '--one', lib, shape])
