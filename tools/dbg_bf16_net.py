import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
from synthsr_amd.unet import unet
from oracle import unet_ref as U
def run(feats, levels, shape, cin, dtype):
    net = unet(nb_features=feats, input_shape=list(shape) + [cin], nb_levels=levels, conv_size=3, nb_labels=1,
               feat_mult=2, nb_conv_per_level=2, final_pred_activation='linear', batch_norm=-1, activation='elu', seed=3, dtype=dtype, fold_upsample=False)
    g = torch.Generator().manual_seed(11)
    for nm, v in net.named_parameters():
        if nm.endswith('/gamma'): v.copy_(torch.rand(v.shape, generator=g) + .5)
        elif nm.endswith('/beta') or nm.endswith('/bias'): v.copy_(torch.randn(v.shape, generator=g) * .1)
    net.repack()
    x = torch.rand(*shape, cin, generator=g); target = torch.rand(*shape, 1, generator=g)
    loss, pred = net.loss_l1(x.cuda(), target.reshape(-1).cuda(), want_pred=True)
    net.backward()
    P = {nm: v.detach().cpu().clone().requires_grad_(True) for nm, v in net.named_parameters()}
    pr = U.unet_forward(x, P, net.prefix, levels, 2, training=True)
    lr = U.l1_loss(pr, target); lr.backward()
    print(dtype, 'levels', levels, 'loss', loss.item(), float(lr))
    for nm, _, kind in net.specs:
        got = net.view(nm, net.grads).cpu().double().reshape(-1); ref = P[nm].grad.double().reshape(-1)
        cos = float(torch.dot(got, ref) / (got.norm() * ref.norm()).clamp_min(1e-30))
        print('  %-32s err %.3e cos %.5f' % (nm, float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)), cos))
run(24, 1, (16,16,32), 2, 'bf16')
run(24, 2, (16,16,32), 2, 'bf16')
