import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synthsr_amd import ops, _lib
lib = _lib.load()
for D, ci, co in [(64, 24, 24), (80, 48, 48), (160, 24, 24)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(D, D, D, ci, generator=g).cuda(); w = (torch.randn(3, 3, 3, ci, co, generator=g) * .05).cuda()
    b = torch.randn(co, generator=g).cuda()
    for bias in (b, None):
        for act in (0, 1):
            outs = []
            for persist in (0, 1):
                lib.synthsr_conv3d_set_option(0, persist)
                wp = ops.pack_conv_weights(w, (D, D, D), 0)
                outs.append(ops.conv3d(x, wp, bias, co, act).clone())
            d = (outs[0] - outs[1]).abs().max().item()
            print(D, ci, co, 'bias' if bias is not None else 'nobias', 'act', act, 'maxdiff', d, 'finite', torch.isfinite(outs[1]).all().item(),
                  'nan count', torch.isnan(outs[1]).sum().item())
