import sys, os
sys.path.insert(0, '/root/repo')
import torch
from synthsr_amd import ops, _lib
from tools.conv_bench import t
lib = _lib.load()
D=160
x = torch.randn(D, D, D, 2, device='cuda'); dy = torch.randn(D, D, D, 24, device='cuda'); dw = torch.zeros(3,3,3,2,24, device='cuda')
for gx in [1024, 512, 256, 128, 64]:
    lib.synthsr_conv3d_set_option(2, gx)
    ms = min(t(lambda: ops.conv3d_wgrad(x, dy, dw), 10) for _ in range(3))
    print(gx, '%.3f ms' % ms)
lib.synthsr_conv3d_set_option(2, 0)
