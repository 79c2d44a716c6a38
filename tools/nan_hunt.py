import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from synthsr_amd import ops, _lib
from synthsr_amd.unet import unet
lib = _lib.load()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 160
for persist in (0, 1):
    lib.synthsr_conv3d_set_option(0, persist)
    net = unet(24, [S, S, S, 2], 5, 3, 1, feat_mult=2, nb_conv_per_level=2, final_pred_activation='linear', batch_norm=-1, seed=0)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(S, S, S, 2, generator=g).cuda(); tgt = torch.rand(S * S * S, generator=g).cuda()
    for step in range(3):
        loss, _ = net.loss_l1(x, tgt)
        bad = []
        for l, acts in enumerate(net.saved['enc']):
            for k, a in enumerate(acts):
                if not torch.isfinite(a).all(): bad.append('enc%d_%d' % (l, k))
        for l, acts in enumerate(net.saved['dec']):
            for k, a in enumerate(acts):
                if not torch.isfinite(a).all(): bad.append('dec%d_%d' % (l, k))
        net.backward()
        gbad = [nm for nm, _, _ in net.specs if not torch.isfinite(net.view(nm, net.grads)).all()]
        print('persist', persist, 'step', step, 'loss', loss.item(), 'bad acts', bad[:6], 'bad grads', gbad[:6])
        net.adam_step(1e-4); net.update_moving_stats()
