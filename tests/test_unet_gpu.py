"""GPU parity of the U-Net kernels (through the C ABI) against the plain PyTorch-CPU float32 restatement
(oracle/unet_ref.py).  Tolerances: conv activations / gradients rel 2e-4 of the tensor's max-abs
(float32 accumulation-order differences over K = 27*Cin up to 15552 and over up to 4e6 voxels);
pointwise kernels 1e-5."""
import numpy as np
import pytest

from conftest import single_shot_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def T():
    import torch
    assert torch.cuda.is_available()
    torch.manual_seed(0)
    return torch


def close(a, b, rel=2e-4, name=''):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(b.abs().max().item(), 1e-30)
    err = (a - b).abs().max().item() / scale
    assert err < rel, '%s: max rel err %.3e (scale %.3e)' % (name, err, scale)


CONV_CASES = [  # (shape, Cin, Cout)
    ((8, 8, 16), 24, 24), ((9, 7, 21), 24, 48), ((5, 6, 18), 48, 24), ((4, 4, 16), 72, 24), ((6, 5, 17), 2, 24),
    ((6, 5, 17), 1, 24), ((4, 8, 16), 96, 96), ((3, 4, 5), 192, 384), ((4, 4, 4), 576, 192), ((10, 10, 10), 24, 1 * 16),
    ((12, 12, 12), 8, 16), ((6, 6, 33), 144, 48), ((10, 10, 10), 384, 384), ((20, 20, 20), 96, 192),
    ((40, 40, 40), 48, 96), ((48, 40, 64), 24, 24), ((32, 48, 64), 72, 24),
    # large enough (>= 768 tiles) for the persistent forward kernel: 1, 2 and 3 input-channel chunks, ragged edges
    ((64, 64, 64), 24, 24), ((40, 64, 80), 48, 48), ((32, 64, 96), 72, 24), ((61, 50, 70), 48, 24),
]


@pytest.mark.parametrize('shape,Cin,Cout', CONV_CASES)
def test_conv3d_fwd_dgrad_wgrad(T, shape, Cin, Cout):
    torch = T
    from synthsr_amd import ops
    from oracle import unet_ref as U
    g = torch.Generator().manual_seed(Cin * 1000 + Cout)
    x = torch.randn(*shape, Cin, generator=g)
    w = torch.randn(3, 3, 3, Cin, Cout, generator=g) / np.sqrt(27 * Cin)
    b = torch.randn(Cout, generator=g)
    dy = torch.randn(*shape, Cout, generator=g)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yr = U.conv3d_same(xr, wr, b)
    yr_elu = torch.nn.functional.elu(yr)
    yr.backward(dy)
    xd, wd, bd, dyd = x.cuda(), w.cuda(), b.cuda(), dy.cuda()
    wp = ops.pack_conv_weights(wd, shape, 0)
    close(ops.conv3d(xd, wp, bd, Cout, act=0), yr, name='fwd linear')
    close(ops.conv3d(xd, wp, bd, Cout, act=1), yr_elu, name='fwd elu')
    wpd = ops.pack_conv_weights(wd, shape, 1)
    close(ops.conv3d(dyd, wpd, None, Cin, act=0), xr.grad, name='dgrad')
    dw = torch.zeros_like(wd)
    ops.conv3d_wgrad(xd, dyd, dw)
    close(dw, wr.grad, name='wgrad')
    # dbias from the constant-1 row of the weight-gradient GEMM
    dw2, db = torch.zeros_like(wd), torch.zeros(Cout, device='cuda')
    ops.conv3d_wgrad(xd, dyd, dw2, dbias=db)
    close(dw2, wr.grad, name='wgrad (with dbias)')
    close(db, dy.reshape(-1, Cout).sum(0), name='dbias')
    # data gradient fused with the ELU backward of the layer below (y = that layer's ELU output)
    ybelow = torch.nn.functional.elu(torch.randn(*shape, Cin, generator=g))
    deriv = torch.where(ybelow > 0, torch.ones_like(ybelow), ybelow + 1)
    close(ops.conv3d_add(dyd, wpd, None, ybelow.cuda(), Cin, act=2), xr.grad * deriv, name='dgrad * elu\'')


def test_conv3d_identity_and_transpose_detecting(T):
    """asymmetric one-hot kernels: catches a swapped tap / channel mapping that random data could average out"""
    torch = T
    from synthsr_amd import ops
    x = torch.randn(6, 7, 19, 24)
    for tap, ci, co in [((0, 1, 2), 3, 17), ((2, 0, 1), 23, 0), ((1, 1, 1), 5, 5)]:
        w = torch.zeros(3, 3, 3, 24, 24)
        w[tap[0], tap[1], tap[2], ci, co] = 1.0
        y = ops.conv3d(x.cuda(), ops.pack_conv_weights(w.cuda(), (6, 7, 19), 0), None, 24, act=0).cpu()
        exp = torch.zeros(6, 7, 19)
        xp = torch.nn.functional.pad(x[..., ci], (1, 1, 1, 1, 1, 1))
        exp = xp[tap[0]:tap[0] + 6, tap[1]:tap[1] + 7, tap[2]:tap[2] + 19]
        assert torch.equal(y[..., co], exp)
        assert int((y != 0).sum()) == int((exp != 0).sum())  # nothing leaks into other channels / voxels


@pytest.mark.parametrize('shape', [(6, 7, 19), (80, 80, 80)])  # plain and persistent kernels
def test_elu_epilogue_accuracy(T, shape):
    """the fused ELU (exp2-based, csrc/conv3d.hip elu_f) against expm1 in float64 through an identity convolution:
    tolerance 2e-7 absolute, 2e-6 relative"""
    torch = T
    from synthsr_amd import ops
    n = shape[0] * shape[1] * shape[2] * 24
    x = torch.cat([-torch.logspace(-7, 1.2, n // 2), torch.randn(n - n // 2) * 3])[torch.randperm(n)].reshape(*shape, 24)
    w = torch.zeros(3, 3, 3, 24, 24)
    w[1, 1, 1] = torch.eye(24)
    y = ops.conv3d(x.cuda(), ops.pack_conv_weights(w.cuda(), shape, 0), None, 24, act=1).cpu().double()
    xd = x.double()
    ref = torch.where(xd > 0, xd, torch.expm1(xd))
    err = (y - ref).abs()
    assert float(err.max()) < 2e-7
    assert float((err / ref.abs().clamp_min(1e-30)).max()) < 2e-6


@pytest.mark.parametrize('lo_shape,Cs,Cl,Cout', [((6, 5, 9), 24, 48, 24), ((4, 4, 8), 48, 96, 48), ((3, 2, 3), 24, 24, 48),
                                                 ((20, 20, 24), 24, 48, 24),
                                                 # >= 768 tiles: the all-parity 4x4x1 kernel (Cout = 24), ragged x
                                                 ((64, 48, 56), 24, 48, 24)])
def test_upsample_folded_conv(T, lo_shape, Cs, Cl, Cout):
    """conv on concatenate([skip, UpSampling3D(2)(lo)]) evaluated as conv3(skip) + 8 parity convs on lo: forward,
    both data gradients and the full weight gradient against autograd on the materialised concat"""
    torch = T
    from synthsr_amd import ops
    from oracle import unet_ref as U
    g = torch.Generator().manual_seed(Cs + 7 * Cl)
    full = tuple(2 * s for s in lo_shape)
    skip = torch.randn(*full, Cs, generator=g)
    lo = torch.randn(*lo_shape, Cl, generator=g)
    w = torch.randn(3, 3, 3, Cs + Cl, Cout, generator=g) / np.sqrt(27 * (Cs + Cl))
    b = torch.randn(Cout, generator=g)
    dy = torch.randn(*full, Cout, generator=g)
    sr, lr, wr = skip.clone().requires_grad_(True), lo.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = torch.nn.functional.elu(U.conv3d_same(torch.cat([sr, U.upsample2(lr)], -1), wr, b))
    # backward through the pre-activation to compare with our dz-based kernels: dz = dy * elu'(y)
    yr.backward(dy)
    dz = dy * torch.where(yr.detach() > 0, torch.ones_like(dy), yr.detach() + 1)
    wd, sd, ld, dzd = w.cuda(), skip.cuda(), lo.cuda(), dz.cuda()
    wp_s = ops.pack_conv_weights_ex(wd, full, 0, Cs, 0, False)
    wp_u = ops.pack_conv_weights_ex(wd, lo_shape, Cs, Cl, 0, True)
    tmp = ops.conv3d(sd, wp_s, None, Cout, act=0)
    y = ops.conv3d_up(ld, wp_u, b.cuda(), tmp, Cout, act=1)
    close(y, yr, name='folded forward')
    wpd_s = ops.pack_conv_weights_ex(wd, full, 0, Cs, 1, False)
    wpd_u = ops.pack_conv_weights_ex(wd, lo_shape, Cs, Cl, 1, True)
    close(ops.conv3d(dzd, wpd_s, None, Cs, act=0), sr.grad, name='dskip')
    close(ops.conv3d_up_dgrad(dzd, wpd_u, Cl), lr.grad, name='dlo')
    dw = torch.zeros_like(wd)
    ops.conv3d_wgrad_part(sd, dzd, dw, 0)
    dwc = torch.empty(8, 27, Cl, Cout, device='cuda')
    ops.conv3d_up_wgrad(ld, dzd, dwc, dw, Cs)
    close(dw, wr.grad, name='dW')


def test_pointwise_kernels(T):
    torch = T
    from synthsr_amd import ops
    from oracle import unet_ref as U
    shape, C = (8, 6, 10), 24
    x = torch.randn(*shape, C) * 2 + 0.5
    gamma, beta = torch.rand(C) + .5, torch.randn(C)
    gamma[3] = -0.7  # negative scale: max-pool must be taken AFTER the BN
    xd = x.cuda()
    stats = torch.zeros(2 * C, device='cuda')
    ws = torch.zeros(2 * C, dtype=torch.float64, device='cuda')
    ops.bn_stats(xd, stats, ws)
    yr, m, v = U.batchnorm_train(x, gamma, beta)
    close(stats[:C], m, 1e-5, 'mean')
    close(stats[C:], v, 1e-5, 'var')
    close(ops.bn_apply(xd, stats, gamma.cuda(), beta.cuda()), yr, 1e-5, 'bn_apply')
    close(ops.bn_maxpool(xd, stats, gamma.cuda(), beta.cuda()), U.maxpool2(yr), 1e-5, 'bn_maxpool')
    # backward of BN + maxpool against autograd
    xr = x.clone().requires_grad_(True)
    y2, _, _ = U.batchnorm_train(xr, gamma, beta)
    p = U.maxpool2(y2)
    dp = torch.randn_like(p)
    p.backward(dp)
    gbn = ops.bn_maxpool_bwd(dp.cuda(), xd, stats, gamma.cuda(), beta.cuda())
    sums = torch.zeros(2 * C, device='cuda')
    dx = ops.bn_bwd(gbn, xd, stats, gamma.cuda(), sums)
    close(dx, xr.grad, 2e-4, 'bn+pool backward')
    # the pool backward can emit the BN-backward sums itself; the head's rank-1 gradient likewise
    sums2 = torch.zeros(2 * C, device='cuda')
    gbn2 = ops.bn_maxpool_bwd(dp.cuda(), xd, stats, gamma.cuda(), beta.cuda(), sums=sums2)
    assert torch.equal(gbn2, gbn)
    close(sums2, sums, 1e-4, 'BN sums from the pool backward')
    dpred, wh = torch.randn(*shape), torch.randn(C)
    dbn = torch.empty_like(xd)
    dw_a, db_a = torch.zeros(C, device='cuda'), torch.zeros(1, device='cuda')
    ops.head_bwd(dpred.cuda(), xd, stats, gamma.cuda(), beta.cuda(), wh.cuda(), dbn, dw_a, db_a)
    close(dbn, dpred[..., None] * wh, 1e-6, 'head dbn')
    close(dw_a, (dpred[..., None] * yr).reshape(-1, C).sum(0), 1e-4, 'head dw')
    s_ref = torch.zeros(2 * C, device='cuda')
    ops.bn_reduce_bwd(dbn, xd, stats, s_ref)
    dw_b, db_b, s_b = torch.zeros(C, device='cuda'), torch.zeros(1, device='cuda'), torch.zeros(2 * C, device='cuda')
    ops.head_bwd(dpred.cuda(), xd, stats, gamma.cuda(), beta.cuda(), wh.cuda(), None, dw_b, db_b, bn_sums=s_b)
    close(dw_b, dw_a, 1e-5, 'head dw (no dbn)')
    close(db_b, db_a, 1e-5, 'head db (no dbn)')
    close(s_b, s_ref, 1e-4, 'BN sums from head_bwd')
    yelu = torch.nn.functional.elu(x).cuda()
    close(ops.bn_elu_bwd_head(dpred.cuda(), wh.cuda(), yelu, stats, gamma.cuda(), s_ref),
          ops.bn_elu_bwd(dbn, yelu, stats, gamma.cuda(), s_ref), 1e-6, 'rank-1 bn_elu_bwd')
    # ELU backward + bias gradient
    y = torch.nn.functional.elu(x)
    dy, dy2 = torch.randn_like(y), torch.randn_like(y)
    dbias = torch.zeros(C, device='cuda')
    dz = ops.elu_bwd(dy.cuda(), y.cuda(), dy2=dy2.cuda(), dbias=dbias)
    ref = (dy + dy2) * torch.where(x > 0, torch.ones_like(x), torch.exp(x))
    close(dz, ref, 1e-5, 'elu_bwd')
    close(dbias, ref.reshape(-1, C).sum(0), 1e-4, 'dbias')
    # upsample + concat and its backward
    lo = torch.randn(4, 3, 5, 48)
    st_lo = torch.zeros(96, device='cuda')
    ws2 = torch.zeros(96, dtype=torch.float64, device='cuda')
    ops.bn_stats(lo.cuda(), st_lo, ws2)
    g2, b2 = torch.rand(48) + .5, torch.randn(48)
    cat = ops.upsample_concat(xd, lo.cuda(), st_lo, g2.cuda(), b2.cuda())
    lo_bn, _, _ = U.batchnorm_train(lo, g2, b2)
    close(cat, torch.cat([x, U.upsample2(lo_bn)], -1), 1e-5, 'upsample_concat')
    dcat = torch.randn(*shape, C + 48)
    dskip, dlo = ops.upsample_concat_bwd(dcat.cuda(), C, 48)
    close(dskip, dcat[..., :C], 1e-6, 'dskip')
    ref_dlo = dcat[..., C:].reshape(4, 2, 3, 2, 5, 2, 48).sum((1, 3, 5))
    close(dlo, ref_dlo, 1e-5, 'dlo')


def _copy_params(net, torch):
    return {nm: v.detach().cpu().clone().requires_grad_(True) for nm, v in net.named_parameters()}


@pytest.mark.parametrize('fold', [False, True])
@pytest.mark.parametrize('feats,levels,shape,cin', [(24, 3, (16, 16, 32), 2), (8, 2, (8, 12, 16), 1), (24, 5, (32, 32, 32), 2)])
def test_unet_loss_and_gradients_vs_autograd(T, feats, levels, shape, cin, fold):
    """one step of the whole network against the oracle under autograd; single shot in deterministic mode, then the default
    atomics path with any max-pool tie flip identified (conftest.single_shot_parity)"""
    torch = T
    from synthsr_amd.unet import unet
    from oracle import unet_ref as U
    g = torch.Generator().manual_seed(11)
    tensors = {}

    def run():
        net = unet(nb_features=feats, input_shape=list(shape) + [cin], nb_levels=levels, conv_size=3, nb_labels=1,
                   feat_mult=2, nb_conv_per_level=2, final_pred_activation='linear', batch_norm=-1, activation='elu', seed=3,
                   fold_upsample=fold)
        # make BN affine and biases non-trivial
        g.manual_seed(11)
        for nm, v in net.named_parameters():
            if nm.endswith('/gamma'):
                v.copy_(torch.rand(v.shape, generator=g) + .5)
            elif nm.endswith('/beta') or nm.endswith('/bias'):
                v.copy_(torch.randn(v.shape, generator=g) * .1)
        net.repack()
        tensors['x'] = torch.rand(*shape, cin, generator=g)
        tensors['target'] = torch.rand(*shape, 1, generator=g)
        loss, pred = net.loss_l1(tensors['x'].cuda(), tensors['target'].reshape(-1).cuda(), want_pred=True)
        net.test_loss, net.test_pred = loss.clone(), pred.clone()
        net.backward()
        return net

    def oracle(net, nudge):
        P = _copy_params(net, torch)
        stats, pin = {}, []
        pr = U.unet_forward(tensors['x'], P, net.prefix, levels, 2, training=True, collect=stats, pool_inputs=pin,
                            pool_nudge=nudge)
        lr = U.l1_loss(pr, tensors['target'])
        lr.backward()
        return (P, stats, pr.detach(), lr.detach()), pin

    def compare(net, ref):
        P, stats, pr, lr = ref
        close(net.test_pred.view(*shape, 1), pr, 5e-4, 'prediction')
        assert abs(net.test_loss.item() - lr.item()) < 2e-5 * max(1.0, abs(lr.item()))
        # (every parameter gradient: the float64-anchored rule of conftest.single_shot_parity)
        # batch statistics
        for bn in net.bn_layers:
            o, C = bn['soff'], bn['C']
            close(net.bn_batch[o:o + C], stats[bn['name']][0], 1e-4, bn['name'] + ' mean')
            close(net.bn_batch[o + C:o + 2 * C], stats[bn['name']][1], 1e-4, bn['name'] + ' var')

    net, _ = single_shot_parity(run, oracle, compare)
    x = tensors['x']
    # one Keras-Adam step
    p0 = net.params.clone()
    g0 = net.grads.clone()
    net.adam_step(lr=1e-3)
    pref, _, _ = U.adam_keras(p0.cpu(), g0.cpu(), torch.zeros_like(p0.cpu()), torch.zeros_like(p0.cpu()), 1, lr=1e-3)
    close(net.params, pref, 1e-6, 'adam')
    # inference path with moving statistics runs and is finite
    net.update_moving_stats()
    out = net.predict(x.cuda())
    assert torch.isfinite(out).all() and list(out.shape) == list(shape) + [1]


def test_training_reduces_loss(T):
    """a few steps of the full loop (generator -> U-Net -> Adam) on a tiny volume: loss is finite and decreases on a
    fixed sample"""
    torch = T
    from synthsr_amd.brain_generator import BrainGenerator
    from synthsr_amd.training import Trainer
    from synthsr_amd.unet import unet
    from synthsr_amd.synthetic import (synthetic_label_pool, GENERATION_LABELS, GENERATION_CLASSES, PRIOR_MEANS_T1_HR,
                                       PRIOR_STDS_T1_HR)
    pool = synthetic_label_pool(2, (32, 32, 32), 5)
    bg = BrainGenerator(None, PRIOR_MEANS_T1_HR, PRIOR_STDS_T1_HR, 'normal', GENERATION_LABELS,
                        generation_classes=GENERATION_CLASSES, output_shape=32, output_div_by_n=8, nonlin_std=4.,
                        nonlin_shape_factor=.125, bias_shape_factor=.125, build_reliability_maps=True, downsample=True,
                        shearing_bounds=.02, label_maps=pool, rng=np.random.default_rng(0))
    net = unet(24, bg.model_output_shape, 3, 3, 1, feat_mult=2, nb_conv_per_level=2, final_pred_activation='linear',
               batch_norm=-1, seed=1)
    tr = Trainer(bg, net, lr=1e-3)
    inputs = next(bg.model_inputs_generator)
    draws = bg.labels_to_image_model.sample_draws()
    losses = [tr.step(inputs, draws).item() for _ in range(8)]
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0]


def test_segmentation_loss_with_the_laplace_head_vs_autograd(T):
    """regression_metric='laplace' together with the segmentation-regularised loss (one regression target: a 2-channel head,
    SynthSR/metrics_model.py:33-49): `predicted_image` -- what the frozen segmentation network sees -- is the INTENSITY
    channel (:53), so the Dice gradient lands on head channel 0 only.  Loss values and every gradient of the trained
    network against autograd through the oracle (single shot, conftest.single_shot_parity)."""
    torch = T
    from synthsr_amd.unet import unet
    from synthsr_amd.seg_loss import SegmentationRegulariser
    from oracle import unet_ref as U
    shape, levels, w = (16, 24, 32), 3, 0.25
    gen_labels = np.array([0, 14, 2, 3, 41, 42, 17])
    seg_labels = np.array([0, 2, 3, 4, 41, 42, 43, 17, 53])
    equivalency = np.array([0, 2, 3, 3, 41, 42, 42, 17, 17])
    g = torch.Generator().manual_seed(5)
    x = torch.rand(*shape, 2, generator=g)
    target = torch.rand(*shape, generator=g)
    seg_target = torch.randint(0, len(gen_labels), shape, generator=g, dtype=torch.int32)

    def nets():
        net = unet(24, list(shape) + [2], levels, 3, 2, feat_mult=2, nb_conv_per_level=2, batch_norm=-1, activation='elu',
                   final_pred_activation='linear', seed=3)
        segnet = unet(24, list(shape) + [1], levels, 3, len(seg_labels), feat_mult=2, nb_conv_per_level=2, batch_norm=-1,
                      activation='elu', final_pred_activation='softmax', seed=4)
        return net, segnet

    def run():
        net, segnet = nets()
        reg = SegmentationRegulariser(segnet, gen_labels, equivalency, w)
        loss, pred = net.loss(x.cuda(), target.cuda().reshape(-1), 'laplace', want_pred=True)
        net.test_loss = loss.clone()
        net.test_dice = reg(pred, seg_target.cuda(), net.dpred, None, head_channels=2).clone()
        net.backward()
        net.test_segnet = segnet
        return net

    def oracle(net, nudge):
        P = {k: v.clone().float().requires_grad_(True) for k, v in net.state_dict().items() if 'moving' not in k}
        Pseg = {k: v.clone().float() for k, v in net.test_segnet.state_dict().items()}
        pin, pin_seg, n1 = [], [], levels - 1
        pr = U.unet_forward(x, P, net.prefix, levels, 2, training=True, pool_inputs=pin,
                            pool_nudge=None if nudge is None else nudge[:n1])
        lap = U.regression_loss(pr, target[..., None], 'laplace')
        dref = U.seg_regularisation(pr[..., 0], seg_target, Pseg, net.test_segnet.prefix, levels, 2, gen_labels, equivalency,
                                    pool_inputs=pin_seg, pool_nudge=None if nudge is None else nudge[n1:],
                                    bn_batch_stats=True)
        (lap + w * dref).backward()
        return (P, lap.detach(), dref.detach()), pin + pin_seg

    def compare(net, ref):
        P, lap, dref = ref
        assert abs(float(net.test_loss.item()) - float(lap)) < 2e-5 * max(1.0, abs(float(lap)))
        assert abs(float(net.test_dice.item()) - float(dref)) < 2e-5
        # (every parameter gradient: the float64-anchored rule of conftest.single_shot_parity)

    # (max_flips 32: see test_segmentation_regularised_loss_vs_autograd)
    single_shot_parity(run, oracle, compare, loss_of=lambda n_: n_.test_loss, pool_nets=lambda n_: [n_, n_.test_segnet], max_flips=32)
    # two regression targets: refused like the reference's graph (the segmentation network takes ONE channel)
    from synthsr_amd.training import Trainer
    net4 = unet(24, list(shape) + [2], levels, 3, 4, feat_mult=2, nb_conv_per_level=2, batch_norm=-1, activation='elu',
                final_pred_activation='linear', seed=3)

    class _BG:
        labels_to_image_model = None
    with pytest.raises(ValueError):
        Trainer(_BG(), net4, regression_metric='laplace', seg_regulariser=object())


@pytest.mark.parametrize('fs_header,clip,crop,frozen_bn,seg_drop', [
    (False, False, None, 'batch', 0.), (True, True, None, 'inference', 0.), (True, False, (12, 16, 20), 'batch', 0.),
    (False, True, (8, 24, 12), 'inference', 0.), (True, True, (8, 24, 12), 'batch', 0.),
    # the frozen network built with conv_dropout (SynthSR/training.py:381): its Dropout layers are active in the learning phase
    (False, False, None, 'batch', .3), (True, True, (8, 24, 12), 'batch', .3)])
def test_segmentation_regularised_loss_vs_autograd(T, fs_header, clip, crop, frozen_bn, seg_drop):
    """SynthSR/metrics_model.py:136-215: L1 + w * Dice(frozen segmentation U-Net(prediction), label map).  Loss value and
    every gradient of the TRAINED network against torch autograd through the oracle; the frozen network's BatchNorm on batch
    statistics (Keras' learning phase, the default) or on its moving averages.
    Single shot in deterministic mode; on the atomics path the pooling choices of BOTH networks are compared
    (conftest.single_shot_parity)"""
    torch = T
    from synthsr_amd.unet import unet
    from synthsr_amd.seg_loss import SegmentationRegulariser
    from oracle import unet_ref as U
    shape, levels = (16, 24, 32), 3
    gen_labels = np.array([0, 14, 2, 3, 41, 42, 17])
    seg_labels = np.array([0, 2, 3, 4, 41, 42, 43, 17, 53])          # labels the segmentation net predicts
    equivalency = np.array([0, 2, 3, 3, 41, 42, 42, 17, 17])        # ... mapped onto generation-label VALUES (merges)
    segshape = (shape[0], shape[2], shape[1]) if fs_header else shape
    g = torch.Generator().manual_seed(0)
    moving = None
    m, M = (0.1, 0.7) if clip else (None, None)
    w = 0.25

    def nets():
        net = unet(24, list(shape) + [2], levels, 3, 1, feat_mult=2, nb_conv_per_level=2, batch_norm=-1, activation='elu',
                   final_pred_activation='linear', seed=3)
        segnet = unet(24, list(segshape) + [1], levels, 3, len(seg_labels), feat_mult=2, nb_conv_per_level=2, batch_norm=-1,
                      activation='elu', final_pred_activation='softmax', seed=4, conv_dropout=seg_drop)
        return net, segnet

    net, segnet = nets()
    seg_scales = None
    if seg_drop:
        rng = np.random.default_rng(21)
        seg_scales = {}
        for c in segnet.all_convs():
            keep = rng.random(c['cout']) >= seg_drop
            keep[:2] = [False, True]
            seg_scales[c['name']] = (keep / (1.0 - seg_drop)).astype(np.float32)
    moving = torch.rand(segnet.bn_moving.shape, generator=g) * 0.5 + 0.25
    x = torch.rand(*shape, 2, generator=g)
    target = torch.rand(*shape, generator=g)
    seg_target = torch.randint(0, len(gen_labels), shape, generator=g, dtype=torch.int32)  # indices, cf. the module docstring
    del net, segnet

    def make_run(rel_weight, dice_only):
        def run():
            net, segnet = nets()
            segnet.bn_moving.copy_(moving.to(segnet.device))
            reg = SegmentationRegulariser(segnet, gen_labels, equivalency, rel_weight, m=m, M=M, fs_header=fs_header,
                                          frozen_bn=frozen_bn)
            loss, pred = net.loss_l1(x.cuda(), target.cuda().reshape(-1), want_pred=True)
            if dice_only:  # image-loss gradient zeroed
                net.dpred.zero_()
            net.test_loss = loss.clone()
            if seg_scales is not None:
                segnet.set_dropout_scales(seg_scales)
            net.test_dice = reg(pred, seg_target.cuda(), net.dpred, crop).clone()
            net.backward()
            net.test_segnet = segnet
            return net
        return run

    def make_oracle(dice_only):
        def oracle(net, nudge):          # nudge / pool inputs: the pooled levels of the trained net, then of the frozen one
            P = {k: v.clone().float().requires_grad_(True) for k, v in net.state_dict().items() if 'moving' not in k}
            Pseg = {k: v.clone().float() for k, v in net.test_segnet.state_dict().items()}
            pin, pin_seg = [], []
            n1 = levels - 1
            pr = U.unet_forward(x, P, net.prefix, levels, 2, training=True, pool_inputs=pin,
                                pool_nudge=None if nudge is None else nudge[:n1])[..., 0]
            l1 = (pr - target).abs().mean()
            dref = U.seg_regularisation(pr, seg_target, Pseg, net.test_segnet.prefix, levels, 2, gen_labels, equivalency, m=m,
                                        M=M, fs_header=fs_header, loss_cropping=crop, pool_inputs=pin_seg,
                                        pool_nudge=None if nudge is None else nudge[n1:],
                                        bn_batch_stats=frozen_bn == 'batch',
                                        dropout=None if seg_scales is None else {k: torch.from_numpy(v)
                                                                                 for k, v in seg_scales.items()})
            (dref if dice_only else l1 + w * dref).backward()
            return (P, l1.detach(), dref.detach()), pin + pin_seg
        return oracle

    def compare_total(net, ref):
        P, l1, dref = ref
        assert abs(float(net.test_loss.item()) - float(l1)) < 1e-5
        assert abs(float(net.test_dice.item()) - float(dref)) < 2e-5, (float(net.test_dice.item()), float(dref))
        # (every parameter gradient: the float64-anchored rule of conftest.single_shot_parity)

    def compare_dice(net, ref):
        # the gradients by the float64-anchored rule; the head bias of the Dice term alone is a sum of cancelling contributions
        # (its range is ~0: the rule measures it against 1e-3 of the largest gradient), so also absolutely:
        hb = net.view(net.head['b'], net.grads).cpu()
        assert float((hb - ref[0][net.head['b']].grad).abs().max()) < 1e-5

    both = lambda net: [net, net.test_segnet]
    # max_flips 32: with the segmentation network behind the prediction, d(loss)/d(pred) has more places where a rounding-sized
    # change of the prediction shows grossly (the L1 kinks AND the frozen network's own pooling ties within its receptive field);
    # a soak of the round-5 code (gpurun_out r06p: 6 runs) saw up to 14 such voxels of 16 k, each still <= KINK_ULP apart
    single_shot_parity(make_run(w, False), make_oracle(False), compare_total, pool_nets=both, max_flips=32)
    # the Dice term alone (it is ~1 % of the total gradient here): weight 1
    single_shot_parity(make_run(1.0, True), make_oracle(True), compare_dice, pool_nets=both, max_flips=32)


@pytest.mark.gpu
@pytest.mark.parametrize('kind,crop,with_res,n', [('l1', None, False, 1), ('l1', (8, 6, 12), True, 1), ('l2', None, True, 1),
                                                  ('l2', (10, 12, 4), False, 1), ('laplace', None, False, 1),
                                                  ('laplace', (8, 6, 12), True, 1), ('l1', (8, 6, 12), True, 2),
                                                  ('l2', None, True, 3), ('l1', None, False, 4),
                                                  ('laplace', (8, 6, 12), True, 2)])
def test_head_regression_losses_vs_autograd(T, kind, crop, with_res, n):
    """synthsr_head_loss_fwd / synthsr_head_bwd_multi (metrics_model.py:30-132: l1, l2, laplace, loss_cropping, residual
    channel) against the oracle's regression_loss under autograd.  Tolerance 2e-5 relative (fp32 sums of 3k terms)."""
    torch = T
    from synthsr_amd import ops
    from oracle import unet_ref as U
    g = torch.Generator().manual_seed(21)
    shape, C = (10, 12, 14), 24
    K = 2 * n if kind == 'laplace' else n
    x = torch.randn(*shape, C, generator=g)
    mean, var = torch.randn(C, generator=g) * .1, torch.rand(C, generator=g) + .5
    gamma, beta = torch.rand(C, generator=g) + .5, torch.randn(C, generator=g) * .1
    w = (torch.randn(C, K, generator=g) * .2).requires_grad_(True)
    b = (torch.randn(K, generator=g) * .1).requires_grad_(True)
    target = torch.rand(*shape, n, generator=g)
    image = torch.rand(*shape, 5, generator=g)
    res_ch = [3, 1, 4, 0][:n]
    bn = ((x - mean) * torch.rsqrt(var + ops.BN_EPS) * gamma + beta).requires_grad_(True)
    pred_ref = bn @ w + b
    res = image[..., res_ch] if with_res else None
    loss_ref = U.regression_loss(pred_ref, target, kind, crop, res)
    loss_ref.backward()
    stats = torch.cat([mean, var]).cuda()
    xd = x.cuda()
    loss = torch.zeros(1, device='cuda')
    pred = torch.empty(x[..., 0].numel() * K, device='cuda')
    dpred = torch.empty_like(pred)
    box = None if crop is None else ([int((s - c) / 2) for s, c in zip(shape, crop)], list(crop))
    ops.head_loss_fwd(xd, stats, gamma.cuda(), beta.cuda(), w.detach().cuda(), b.detach().cuda(), target.reshape(-1).cuda(),
                      loss, kind=kind, crop=box, pred=pred, dpred=dpred, residual=image.cuda() if with_res else None,
                      res_stride=5, res_off=res_ch if n > 1 else res_ch[0])
    expect = pred_ref.detach().clone()
    if with_res:
        expect[..., :n] += res
    close(pred.view(*shape, K), expect, 2e-5, 'pred')
    assert abs(loss.item() - loss_ref.item()) < 2e-5 * abs(loss_ref.item())
    # dloss/dpred: recover it from the autograd gradient of bn (dbn = dpred @ w^T) through head_bwd / head_bwd_multi
    dw, db = torch.zeros(C, K, device='cuda'), torch.zeros(K, device='cuda')
    dbn = torch.empty_like(xd)
    if K == 1:
        ops.head_bwd(dpred, xd, stats, gamma.cuda(), beta.cuda(), w.detach().cuda().view(-1), dbn, dw.view(-1), db)
    else:
        ops.head_bwd_multi(dpred, xd, stats, gamma.cuda(), beta.cuda(), w.detach().cuda(), dbn, dw, db)
    close(dbn, bn.grad, 2e-5, 'dbn')
    for got, ref, nm in ((dw, w.grad, 'dw'), (db, b.grad, 'db')):   # sums of +-1/N can cancel to ~0: absolute floor
        err = (got.cpu().double() - ref.double()).abs().max().item()
        assert err < 2e-5 * max(ref.abs().max().item(), 1e-2), '%s abs err %.3e' % (nm, err)
    if crop is not None:  # no gradient outside the box
        d = dpred.view(*shape, K).clone()
        lo, sz = box
        d[lo[0]:lo[0] + sz[0], lo[1]:lo[1] + sz[1], lo[2]:lo[2] + sz[2]] = 0
        assert not d.any()


@pytest.mark.gpu
@pytest.mark.parametrize('kind,crop', [('l2', (8, 8, 16)), ('laplace', None), ('laplace', (16, 8, 8)), ('ssim', None),
                                       ('ssim', (12, 14, 24))])
def test_unet_other_losses_gradients_vs_autograd(T, kind, crop):
    """whole network under regression_metric='l2' / 'laplace' (2-channel head) and loss_cropping vs the oracle"""
    torch = T
    from synthsr_amd.unet import unet
    from oracle import unet_ref as U
    shape, cin, levels = (16, 16, 32), 2, 3
    K = 2 if kind == 'laplace' else 1
    g = torch.Generator().manual_seed(12)
    tensors = {}

    def run():
        net = unet(nb_features=24, input_shape=list(shape) + [cin], nb_levels=levels, conv_size=3, nb_labels=K, feat_mult=2,
                   nb_conv_per_level=2, final_pred_activation='linear', batch_norm=-1, activation='elu', seed=5)
        g.manual_seed(12)
        for nm, v in net.named_parameters():
            if nm.endswith('/gamma'):
                v.copy_(torch.rand(v.shape, generator=g) + .5)
            elif nm.endswith('/beta') or nm.endswith('/bias'):
                v.copy_(torch.randn(v.shape, generator=g) * .1)
        net.repack()
        x = torch.rand(*shape, cin, generator=g)
        target = torch.rand(*shape, 1, generator=g)
        loss, pred = net.loss(x.cuda(), target.reshape(-1).cuda(), kind, crop, residual=x.cuda(), res_stride=cin, res_off=1,
                              want_pred=True)
        net.test_loss, net.test_pred = loss.clone(), pred.clone()
        net.backward()
        tensors.update(x=x, target=target)
        return net

    def oracle(net, nudge):
        P = _copy_params(net, torch)
        pin = []
        pr = U.unet_forward(tensors['x'], P, net.prefix, levels, 2, training=True, pool_inputs=pin, pool_nudge=nudge)
        lr = U.regression_loss(pr, tensors['target'], kind, crop, tensors['x'][..., 1:2])
        lr.backward()
        return (P, pr.detach(), lr.detach()), pin

    def compare(net, ref):
        _, pr, lr = ref
        expect = pr.clone()
        expect[..., :1] += tensors['x'][..., 1:2]
        close(net.test_pred.view(*shape, K), expect, 5e-4, 'prediction')
        assert abs(net.test_loss.item() - lr.item()) < 5e-5 * max(1.0, abs(lr.item()))
        # (every parameter gradient: the float64-anchored rule of conftest.single_shot_parity)

    net, _ = single_shot_parity(run, oracle, compare, loss_of=lambda n_: n_.test_loss)
    x, target = tensors['x'], tensors['target']
    net.update_moving_stats()
    out = net.predict(x.cuda())
    assert torch.isfinite(out).all() and list(out.shape) == list(shape) + [K]
    with pytest.raises(ValueError):
        net.loss(x.cuda(), target.reshape(-1).cuda(), 'l1' if K == 2 else 'laplace')
    if kind == 'ssim':
        return
    with pytest.raises(ValueError):
        net.loss(x.cuda(), target.reshape(-1).cuda(), kind, (16, 16, 40))


@pytest.mark.gpu
@pytest.mark.parametrize('shape,crop', [((14, 16, 13), None), ((20, 24, 30), (12, 16, 14)), ((32, 32, 32), None)])
def test_ssim_loss_and_gradient_vs_autograd(T, shape, crop):
    """regression_metric='ssim' kernels (separable 11-tap passes, forward and transposed) against the oracle's
    tf.image.ssim restatement under autograd.  Tolerance 2e-5 on the loss, 2e-4 of the gradient range (fp32, the oracle
    uses the 121-tap 2-D window, the kernels its two 1-D factors)"""
    torch = T
    from synthsr_amd import ops
    from oracle import unet_ref as U
    g = torch.Generator().manual_seed(31)
    pred = torch.rand(*shape, 1, generator=g).requires_grad_(True)
    target = (0.6 * pred.detach() + 0.4 * torch.rand(*shape, 1, generator=g))
    ref = U.regression_loss(pred, target, 'ssim', crop)
    ref.backward()
    loss = torch.zeros(1, device='cuda')
    dpred = torch.full((pred.numel(),), 7.0, device='cuda')          # must be overwritten, not accumulated into
    box = None if crop is None else ([int((s - c) / 2) for s, c in zip(shape, crop)], list(crop))
    ops.ssim_loss(pred.detach().reshape(-1).cuda(), target.reshape(-1).cuda(), shape, loss, dpred, crop=box)
    assert abs(loss.item() - ref.item()) < 2e-5
    close(dpred.view(*shape, 1), pred.grad, 2e-4, 'dpred')
    # identical volumes: SSIM = 1 -> loss -1, zero gradient
    loss.zero_()
    ops.ssim_loss(target.reshape(-1).cuda(), target.reshape(-1).cuda(), shape, loss, dpred, crop=box)
    assert abs(loss.item() + 1.0) < 1e-5 and dpred.abs().max().item() < 1e-6
    with pytest.raises(ValueError):
        ops.ssim_loss(target.reshape(-1).cuda(), target.reshape(-1).cuda(), shape, loss, dpred, crop=([0, 0, 0], [10, 12, 12]))


def _sample_voxels(shape, n, seed):
    """corners, edges, tile borders of the 4x4x16 tiling and random interior voxels"""
    rng = np.random.RandomState(seed)
    d0, d1, d2 = shape
    pts = [(0, 0, 0), (d0 - 1, d1 - 1, d2 - 1), (0, d1 - 1, 0), (d0 - 1, 0, d2 - 1), (3, 3, 15), (4, 4, 16), (d0 - 4, 3, 16),
           (d0 // 2, d1 // 2, d2 // 2), (d0 - 1, d1 // 2, 15), (7, d1 - 1, d2 - 16)]
    pts += [tuple(int(rng.randint(0, s)) for s in shape) for _ in range(n - len(pts))]
    return np.array(pts)


@pytest.mark.gpu
def test_full_size_160_convs_on_sampled_voxels(T):
    """BASELINE size (160^3, the level-0 layers: 24->24 on the 4x4x1 kernels, 2->24 first layer): forward and data
    gradient checked on 150 voxels (corners, tile borders, random) against the direct float64 sum over the 27 x Cin
    neighbourhood; weight gradient and dbias against float64 GEMMs of the shifted tensors.  Tolerance 2e-5 of the range."""
    torch = T
    from synthsr_amd import ops
    S = (160, 160, 160)
    g = torch.Generator(device='cuda').manual_seed(5)
    vox = _sample_voxels(S, 150, 3)
    for cin in (24, 2):
        x = torch.randn(*S, cin, device='cuda', generator=g)
        w = torch.randn(3, 3, 3, cin, 24, device='cuda', generator=g) * 0.1
        b = torch.randn(24, device='cuda', generator=g)
        y = ops.conv3d(x, ops.pack_conv_weights(w, S, 0), b, 24, act=0)
        xp = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1, 1, 1))
        w64 = w.double().cpu().numpy().reshape(27 * cin, 24)
        ref = np.stack([xp[z:z + 3, yy:yy + 3, xx:xx + 3].double().cpu().numpy().reshape(-1) @ w64
                        for z, yy, xx in vox]) + b.double().cpu().numpy()
        got = y[vox[:, 0], vox[:, 1], vox[:, 2]].double().cpu().numpy()
        assert np.abs(got - ref).max() < 2e-5 * np.abs(ref).max(), (cin, np.abs(got - ref).max())
        del xp
        dy = torch.randn(*S, 24, device='cuda', generator=g)
        if cin == 24:   # data gradient: dx[v][ci] = sum_t sum_co w[t][ci][co] dy[v - (t - 1)][co]
            dx = ops.conv3d(dy, ops.pack_conv_weights(w, S, 1), None, cin, act=0)
            dyp = torch.nn.functional.pad(dy, (0, 0, 1, 1, 1, 1, 1, 1))
            wf = torch.flip(w, dims=[0, 1, 2]).permute(0, 1, 2, 4, 3).double().cpu().numpy().reshape(27 * 24, cin)
            ref = np.stack([dyp[z:z + 3, yy:yy + 3, xx:xx + 3].double().cpu().numpy().reshape(-1) @ wf for z, yy, xx in vox])
            got = dx[vox[:, 0], vox[:, 1], vox[:, 2]].double().cpu().numpy()
            assert np.abs(got - ref).max() < 2e-5 * np.abs(ref).max()
            del dyp, dx
        dw = torch.zeros(3, 3, 3, cin, 24, device='cuda')
        db = torch.zeros(24, device='cuda')
        ops.conv3d_wgrad(x, dy, dw, dbias=db)
        dy2 = dy.reshape(-1, 24).double()
        xp = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1, 1, 1))
        for tz, ty, tx in [(0, 0, 0), (1, 1, 1), (2, 0, 1), (0, 2, 2), (2, 2, 2)]:
            xs = xp[tz:tz + 160, ty:ty + 160, tx:tx + 160].reshape(-1, cin).double()
            ref = (xs.t() @ dy2)
            close(dw[tz, ty, tx], ref, 2e-5, 'dw tap %d%d%d cin %d' % (tz, ty, tx, cin))
            del xs
        close(db, dy2.sum(0), 2e-5, 'dbias')
        del xp, dy2, dy, x, y
        torch.cuda.empty_cache()
