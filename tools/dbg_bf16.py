import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
from synthsr_amd import ops
def rbf(t): return t.bfloat16().float()
torch.manual_seed(0)
x = rbf(torch.randn(6, 7, 19, 24))
for tap, ci, co in [((1,1,1),5,5), ((0, 1, 2), 3, 17), ((2, 0, 1), 23, 0)]:
    w = torch.zeros(3, 3, 3, 24, 24); w[tap[0], tap[1], tap[2], ci, co] = 1.0
    y = ops.conv3d_bf16(x.cuda().bfloat16(), ops.pack_conv_weights_bf16(w.cuda(), 0), None, 24, act=0).float().cpu()
    xp = torch.nn.functional.pad(x[..., ci], (1, 1, 1, 1, 1, 1))
    exp = xp[tap[0]:tap[0] + 6, tap[1]:tap[1] + 7, tap[2]:tap[2] + 19]
    nz = (y != 0).nonzero()
    print(tap, ci, co, 'equal', torch.equal(y[..., co], exp), 'nonzero channels', sorted(set(nz[:,3].tolist())), 'count', len(nz), 'expected', int((exp!=0).sum()))
    d = (y[..., co] - exp)
    bad = (d != 0).nonzero()
    print(' mismatches', len(bad), bad[:10].tolist())
    if len(bad):
        z,yy,xx = bad[0].tolist()
        val = y[z,yy,xx,co].item()
        # where does this value come from in x?
        loc = (x == val).nonzero()
        print(' value', val, 'expected', exp[z,yy,xx].item(), 'found in x at', loc[:5].tolist())
