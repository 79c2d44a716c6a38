#!/usr/bin/env python
"""timing of the head + loss kernel at the benchmark size (160^3 x 24, one l1 target, fused backward sums):
python tools/head_bench.py   (SYNTHSR_HIP_LIB selects the build)"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from synthsr_amd import ops
from conv_bench import t
D, C = 160, 24
g = torch.Generator().manual_seed(3)
x = torch.randn(D, D, D, C, generator=g).cuda()
stats = torch.zeros(2 * C, device='cuda'); ws = torch.zeros(2 * C, dtype=torch.float64, device='cuda')
ops.bn_stats(x, stats, ws)
gamma = (torch.rand(C, generator=g) + .5).cuda(); beta = torch.randn(C, generator=g).cuda()
w = torch.randn(C, 1, generator=g).cuda(); b = torch.randn(1, generator=g).cuda()
tgt = torch.rand(D ** 3, generator=g).cuda(); res = torch.rand(D ** 3, 2, generator=g).cuda()
loss = torch.zeros(1, device='cuda'); pred = torch.empty(D ** 3, device='cuda'); dpred = torch.empty(D ** 3, device='cuda')
ab = torch.zeros(C + 2, device='cuda')


def run():
    loss.zero_(); ab.zero_()
    ops.head_loss_fwd(x, stats, gamma, beta, w, b, tgt, loss, 'l1', None, pred, dpred, res, 2, 1, ab=ab)


ms = min(t(run, 20) for _ in range(3))
torch.cuda.synchronize()
md5 = hashlib.md5(pred.cpu().numpy().tobytes() + dpred.cpu().numpy().tobytes()).hexdigest()[:12]
print('head_loss_fwd 160^3 x 24: %.4f ms (incl. two 4-byte fills)  %.2f TB/s  loss %.7f  pred/dpred md5 %s' % (ms, x.numel() * 4 / 1e9 / ms, loss.item(), md5))
