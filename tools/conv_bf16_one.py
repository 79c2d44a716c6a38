#!/usr/bin/env python
"""a few launches of the bf16 conv kernels on one layer shape (for rocprofv3 --pmc passes):
    python tools/conv_bf16_one.py D Cin Cout"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthsr_amd import ops
D, ci, co = (int(v) for v in sys.argv[1:4])
x = torch.randn(D, D, D, ci, device='cuda').bfloat16()
dz = torch.randn(D, D, D, co, device='cuda').bfloat16()
w = torch.randn(3, 3, 3, ci, co, device='cuda') * .05
b = torch.zeros(co, device='cuda')
wp = ops.pack_conv_weights_bf16(w, 0)
out = torch.empty(D, D, D, co, device='cuda', dtype=torch.bfloat16)
dw = torch.zeros(3, 3, 3, ci, co, device='cuda')
for _ in range(4):
    ops.conv3d_bf16(x, wp, b, co, 1, out=out)
    ops.conv3d_wgrad_bf16(x, dz, dw)
torch.cuda.synchronize()
