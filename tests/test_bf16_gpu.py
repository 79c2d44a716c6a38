"""GPU parity of the bf16 kernels (BASELINE.json configs[3] / [4]; csrc/conv_bf16.hip, bf16 pointwise kernels) through the
C ABI.  The reference is fp32 Keras, so parity is against the fp32 oracle (PyTorch-CPU) evaluated ON THE SAME
bf16-ROUNDED inputs / weights.  Stated bf16 tolerances: a conv output is one bf16 rounding (2^-9 relative) away from
the fp32-accumulated result -> 1e-2 of the tensor's range; fp32 weight gradients (no output rounding) 2e-3 of range."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def T():
    import torch
    assert torch.cuda.is_available()
    return torch


def close(a, b, rel, name=''):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(b.abs().max().item(), 1e-30)
    err = (a - b).abs().max().item() / scale
    assert err < rel, '%s: max rel err %.3e (scale %.3e)' % (name, err, scale)


def rbf(t):
    return t.bfloat16().float()


CASES = [((8, 8, 16), 24, 24), ((9, 7, 21), 24, 48), ((5, 6, 18), 48, 24), ((4, 4, 16), 72, 24), ((6, 5, 17), 8, 24),
         ((4, 8, 16), 96, 96), ((3, 4, 5), 192, 384), ((4, 4, 4), 576, 192), ((12, 12, 12), 32, 64), ((6, 6, 33), 144, 48),
         ((10, 10, 10), 384, 384), ((20, 20, 20), 96, 192), ((48, 40, 64), 24, 24), ((32, 48, 64), 72, 24),
         ((8, 8, 8), 256, 256), ((16, 16, 16), 64, 128)]


@pytest.mark.parametrize('shape,Cin,Cout', CASES)
def test_conv3d_bf16_fwd_dgrad_wgrad(T, shape, Cin, Cout):
    torch = T
    from synthsr_amd import ops
    from oracle import unet_ref as U
    g = torch.Generator().manual_seed(Cin * 1000 + Cout)
    x = rbf(torch.randn(*shape, Cin, generator=g))
    w = rbf(torch.randn(3, 3, 3, Cin, Cout, generator=g) / np.sqrt(27 * Cin))
    b = torch.randn(Cout, generator=g)
    dy = rbf(torch.randn(*shape, Cout, generator=g))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = U.conv3d_same(xr, wr, b)
    yr_elu = torch.nn.functional.elu(yr)
    yr.backward(dy)
    xd, wd, bd, dyd = x.cuda().bfloat16(), w.cuda(), b.cuda(), dy.cuda().bfloat16()
    wp = ops.pack_conv_weights_bf16(wd, 0)
    close(ops.conv3d_bf16(xd, wp, bd, Cout, act=0).float(), yr, 1e-2, 'fwd linear')
    stats = torch.zeros(2 * Cout, device='cuda')
    y = ops.conv3d_bf16(xd, wp, bd, Cout, act=1, stats=stats)
    close(y.float(), yr_elu, 1e-2, 'fwd elu')
    yf = y.float().reshape(-1, Cout)
    close(stats[:Cout], yf.mean(0), 2e-4, 'batch mean (of the rounded output)')
    close(stats[Cout:], yf.var(0, unbiased=False), 2e-4, 'batch variance')
    if Cout % 8 == 0:
        wpd = ops.pack_conv_weights_bf16(wd, 1)
        close(ops.conv3d_bf16(dyd, wpd, None, Cin, act=0).float(), xr.grad, 1e-2, 'dgrad')
        below = rbf(torch.nn.functional.elu(torch.randn(*shape, Cin, generator=g)))
        deriv = torch.where(below > 0, torch.ones_like(below), below + 1)
        close(ops.conv3d_bf16(dyd, wpd, None, Cin, act=2, below=below.cuda().bfloat16()).float(), xr.grad * deriv, 1e-2,
              "dgrad * elu'")
        dw, db = torch.zeros(3, 3, 3, Cin, Cout, device='cuda'), torch.zeros(Cout, device='cuda')
        ops.conv3d_wgrad_bf16(xd, dyd, dw, db)
        close(dw, wr.grad, 2e-3, 'wgrad')
        close(db, dy.reshape(-1, Cout).sum(0), 2e-3, 'dbias')


def test_conv3d_bf16_transpose_detecting(T):
    """asymmetric one-hot kernels: a swapped tap / channel / fragment-lane mapping cannot hide"""
    torch = T
    from synthsr_amd import ops
    x = rbf(torch.randn(6, 7, 19, 24))
    for tap, ci, co in [((0, 1, 2), 3, 17), ((2, 0, 1), 23, 0), ((1, 1, 1), 5, 5), ((2, 2, 0), 8, 23)]:
        w = torch.zeros(3, 3, 3, 24, 24)
        w[tap[0], tap[1], tap[2], ci, co] = 1.0
        y = ops.conv3d_bf16(x.cuda().bfloat16(), ops.pack_conv_weights_bf16(w.cuda(), 0), None, 24, act=0).float().cpu()
        xp = torch.nn.functional.pad(x[..., ci], (1, 1, 1, 1, 1, 1))
        exp = xp[tap[0]:tap[0] + 6, tap[1]:tap[1] + 7, tap[2]:tap[2] + 19]
        assert torch.equal(y[..., co], exp), (tap, ci, co)
        assert int((y != 0).sum()) == int((exp != 0).sum())
    # weight gradient of one-hot dz / x: exactly one (tap, ci, co) entry per shifted overlap
    xs = torch.zeros(6, 7, 19, 24)
    dz = torch.zeros(6, 7, 19, 24)
    xs[2, 3, 5, 7] = 1.0
    dz[3, 3, 4, 13] = 2.0        # x sits at dz position + (-1, 0, +1) -> tap (0, 1, 2)
    dw = torch.zeros(3, 3, 3, 24, 24, device='cuda')
    ops.conv3d_wgrad_bf16(xs.cuda().bfloat16(), dz.cuda().bfloat16(), dw)
    exp = torch.zeros(3, 3, 3, 24, 24)
    exp[0, 1, 2, 7, 13] = 2.0
    assert torch.equal(dw.cpu(), exp)


@pytest.mark.parametrize('lo_shape,Cs,Cl,Cout', [((6, 5, 9), 24, 48, 24), ((4, 4, 8), 48, 96, 48), ((3, 2, 3), 24, 24, 48),
                                                 ((5, 5, 5), 192, 384, 192), ((20, 20, 24), 24, 48, 24),
                                                 ((10, 12, 20), 96, 192, 96), ((40, 40, 48), 24, 48, 24)])
def test_upsample_folded_conv_bf16(T, lo_shape, Cs, Cl, Cout):
    """bf16 nearest-upsample folding (synthsr_conv3d_bf16_up_fwd / _up_dgrad / _up_wgrad, _wgrad_part, the act-5 epilogue):
    conv on concatenate([skip, UpSampling3D(2)(lo)]) = conv3(skip) + 8 parity convs on lo.  Forward, both data gradients
    and the full weight gradient against float32 autograd on the materialised concat of the SAME bf16-rounded operands.
    Tolerances: the up-sampled half's partial sums are rounded to bf16 once before the skip half adds them, and the 8
    parity weight sets are bf16 roundings of SUMS of taps (the unfolded conv sums products of rounded taps): 1.5e-2 of range
    forward / data gradients (1e-2 for the plain bf16 convs), 4e-3 weight gradient (fp32 accumulation of exact bf16 products)."""
    torch = T
    from synthsr_amd import ops
    from oracle import unet_ref as U
    g = torch.Generator().manual_seed(Cs + 7 * Cl)
    full = tuple(2 * s for s in lo_shape)
    skip = rbf(torch.randn(*full, Cs, generator=g))
    lo = rbf(torch.randn(*lo_shape, Cl, generator=g))
    w = rbf(torch.randn(3, 3, 3, Cs + Cl, Cout, generator=g) / np.sqrt(27 * (Cs + Cl)))
    b = torch.randn(Cout, generator=g)
    dy = torch.randn(*full, Cout, generator=g)
    sr, lr, wr = skip.clone().requires_grad_(True), lo.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = torch.nn.functional.elu(U.conv3d_same(torch.cat([sr, U.upsample2(lr)], -1), wr, b))
    dz = rbf(dy * torch.where(yr.detach() > 0, torch.ones_like(dy), yr.detach() + 1))
    # autograd from the (rounded) pre-activation gradient: d/dz of sum(z * dz)
    zr = U.conv3d_same(torch.cat([sr, U.upsample2(lr)], -1), wr, b)
    (zr * dz).sum().backward()
    wd, sd, ld, dzd = w.cuda(), skip.cuda().bfloat16(), lo.cuda().bfloat16(), dz.cuda().bfloat16()
    wp_s = ops.pack_conv_weights_bf16(wd, 0, 0, Cs)
    wp_u = ops.pack_conv_weights_bf16(wd, 0, Cs, Cl, up=True)
    y = ops.conv3d_up(ld, wp_u, None, None, Cout, act=0)
    up_ref = U.conv3d_same(U.upsample2(lo), w[..., Cs:, :])
    close(y.float(), up_ref, 1e-2, 'parity convs (raw partial sums)')
    y = ops.conv3d_add(sd, wp_s, b.cuda(), y, Cout, act=1, out=y)
    close(y.float(), yr, 1.5e-2, 'folded forward')
    wpd_s = ops.pack_conv_weights_bf16(wd, 1, 0, Cs)
    wpd_u = ops.pack_conv_weights_bf16(wd, 1, Cs, Cl, up=True)
    close(ops.conv3d(dzd, wpd_s, None, Cs, act=0).float(), sr.grad, 1e-2, 'dskip')
    close(ops.conv3d_up_dgrad(dzd, wpd_u, Cl).float(), lr.grad, 1.5e-2, 'dlo')
    dw, db = torch.zeros(3, 3, 3, Cs + Cl, Cout, device='cuda'), torch.zeros(Cout, device='cuda')
    ops.conv3d_wgrad_part(sd, dzd, dw, 0, dbias=db)
    dwc = torch.empty(8, 27, Cl, Cout, device='cuda')
    ops.conv3d_up_wgrad(ld, dzd, dwc, dw, Cs)
    close(dw[..., :Cs, :], wr.grad[..., :Cs, :], 4e-3, 'dW (skip channels)')
    close(dw[..., Cs:, :], wr.grad[..., Cs:, :], 4e-3, 'dW (up-sampled channels)')
    close(db, dz.reshape(-1, Cout).sum(0), 4e-3, 'dbias')


def test_upsample_folded_bf16_one_hot(T):
    """asymmetric one-hot kernels through the parity convs: a swapped parity / tap / channel mapping cannot hide (exact)"""
    torch = T
    from synthsr_amd import ops
    from oracle import unet_ref as U
    lo_shape, Cl, Cout = (5, 6, 9), 24, 24
    lo = rbf(torch.randn(*lo_shape, Cl))
    for tap, ci, co in [((0, 1, 2), 3, 17), ((2, 0, 1), 23, 0), ((1, 1, 1), 5, 5), ((2, 2, 0), 8, 23), ((0, 0, 0), 0, 1)]:
        w = torch.zeros(3, 3, 3, Cl, Cout)
        w[tap[0], tap[1], tap[2], ci, co] = 1.0
        y = ops.conv3d_up(lo.cuda().bfloat16(), ops.pack_conv_weights_bf16(w.cuda(), 0, 0, Cl, up=True), None, None, Cout,
                          act=0).float().cpu()
        exp = U.conv3d_same(U.upsample2(lo), w)
        assert torch.equal(y, exp), (tap, ci, co, float((y - exp).abs().max()))


def test_f32_to_bf16_pad(T):
    torch = T
    from synthsr_amd import ops
    x = torch.randn(5, 6, 7, 2)
    y = ops.to_bf16_pad(x.cuda(), 8).float().cpu()
    assert torch.equal(y[..., :2], rbf(x)) and float(y[..., 2:].abs().max()) == 0.0


def _grad_report(net, P):
    """per-tensor (max error / max |ref|, cosine similarity) of the HIP gradients against autograd"""
    import torch
    rep = {}
    for nm, _, kind in net.specs:
        got = net.view(nm, net.grads).cpu().double().reshape(-1)
        ref = P[nm].grad.double().reshape(-1)
        cos = float(torch.dot(got, ref) / (got.norm() * ref.norm()).clamp_min(1e-30))
        rep[nm] = (float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)), cos, kind)
    return rep


@pytest.mark.parametrize('fold', [False, True])
@pytest.mark.parametrize('feats,levels,shape,cin', [(24, 3, (16, 16, 32), 2), (8, 2, (8, 12, 16), 1), (24, 5, (32, 32, 32), 2)])
def test_unet_bf16_step_vs_oracle(T, feats, levels, shape, cin, fold):
    """one training step of the bf16 network (bf16 activations / packed weights, fp32 accumulation, fp32 BatchNorm
    statistics, fp32 master weights and gradients) against the oracle on the same fp32 master weights, two ways:

    (a) the plain fp32 oracle -- what the reference computes.  Stated bf16 tolerances: prediction 3e-2 of its range, loss
        1 %, batch statistics 2e-2 of range.  Gradients only loosely (cosine >= 0.9): max-pooling routes each gradient to
        ONE of 8 voxels and bf16 rounding flips that choice between near-equal candidates (measured: cosine 0.9999 with no
        pooling level above a layer, 0.998 with one, 0.983 with two on iid noise volumes);
    (b) the oracle with the bf16 STORAGE roundings restated (oracle.unet_ref.round_bf16 on the input, conv kernels, conv +
        ELU outputs, pooled and concatenated tensors; everything else fp32): same pooling decisions, so every gradient
        must agree tightly -- cosine >= 0.997, max error <= 8 % of the tensor's range (measured: 0.9995 / 4 %; the residue is
        the occasional activation whose bf16 rounding flips between the two summation orders).  This is the check that pins the
        kernels."""
    torch = T
    from synthsr_amd.unet import unet
    from oracle import unet_ref as U
    net = unet(nb_features=feats, input_shape=list(shape) + [cin], nb_levels=levels, conv_size=3, nb_labels=1,
               feat_mult=2, nb_conv_per_level=2, final_pred_activation='linear', batch_norm=-1, activation='elu', seed=3,
               dtype='bf16', fold_upsample=fold)   # fold: nearest-upsample folding of every decoder level (no concat tensors)
    g = torch.Generator().manual_seed(11)
    for nm, v in net.named_parameters():
        if nm.endswith('/gamma'):
            v.copy_(torch.rand(v.shape, generator=g) + .5)
        elif nm.endswith('/beta') or nm.endswith('/bias'):
            v.copy_(torch.randn(v.shape, generator=g) * .1)
    net.repack()
    x = torch.rand(*shape, cin, generator=g)
    target = torch.rand(*shape, 1, generator=g)
    loss, pred = net.loss_l1(x.cuda(), target.reshape(-1).cuda(), want_pred=True)
    pred = pred.clone()
    net.backward()
    assert net.saved['enc'][0][0].dtype == torch.bfloat16 and net.grads.dtype == torch.float32
    for mode in ('fp32', 'bf16-storage'):
        P = {nm: v.detach().cpu().clone().requires_grad_(True) for nm, v in net.named_parameters()}
        stats = {}
        pr = U.unet_forward(x, P, net.prefix, levels, 2, training=True, collect=stats,
                            quant=None if mode == 'fp32' else U.round_bf16)
        lr = U.l1_loss(pr, target)
        lr.backward()
        tight = mode != 'fp32'
        close(pred.view(*shape, 1), pr, 2.5e-2 if tight else 3e-2, mode + ' prediction')  # one flipped bf16 rounding = 2^-8
        assert abs(loss.item() - lr.item()) < (3e-3 if tight else 1e-2) * abs(lr.item()), (mode, loss.item(), lr.item())
        for bn in net.bn_layers:
            o, C = bn['soff'], bn['C']
            close(net.bn_batch[o:o + C], stats[bn['name']][0], 1.5e-2 if tight else 2e-2, mode + ' ' + bn['name'] + ' mean')
            close(net.bn_batch[o + C:o + 2 * C], stats[bn['name']][1], 1.5e-2 if tight else 2e-2, mode + ' ' + bn['name'] + ' var')
        rep = _grad_report(net, P)
        worst = sorted(((c, e, nm) for nm, (e, c, _) in rep.items()))[:4]
        for nm, (err, cos, kind) in rep.items():
            if tight:   # each pooling level above a layer adds a few flipped arg-max decisions (5 levels at 32^3: 0.991 / 19 %)
                cmin, emax = (0.997, 8e-2) if levels <= 3 else (0.985, 1.0)   # deep nets: cosine only (cancelling sums)
                if fold:   # the folded decoder convs round differently from what `quant` restates (the up-sampled half's
                    # partial sums are rounded to bf16 before the skip half adds them; parity weights = bf16(sum of taps)):
                    # measured 0.9909 / 11 % on the 8-feature 2-level net, 0.996+ on the 24-feature ones
                    cmin, emax = min(cmin, 0.985), max(emax, 0.15)
                assert cos > cmin and err < emax, '%s: gradient of %s: err %.3e cos %.5f (worst %s)' % (mode, nm, err, cos, worst)
            else:
                assert cos > 0.9, '%s: gradient of %s: err %.3e cos %.5f (worst %s)' % (mode, nm, err, cos, worst)
    p0, g0 = net.params.clone(), net.grads.clone()
    net.adam_step(lr=1e-3)
    pref, _, _ = U.adam_keras(p0.cpu(), g0.cpu(), torch.zeros_like(p0.cpu()), torch.zeros_like(p0.cpu()), 1, lr=1e-3)
    close(net.params, pref, 1e-6, 'adam (fp32 master weights)')
    net.update_moving_stats()
    out = net.predict(x.cuda())
    assert torch.isfinite(out).all() and list(out.shape) == list(shape) + [1] and out.dtype == torch.float32


@pytest.mark.parametrize('fold', [False, True])
@pytest.mark.parametrize('B,feats,levels,shape,cin', [(2, 24, 3, (16, 16, 32), 2), (3, 8, 2, (8, 12, 16), 1)])
def test_unet_bf16_per_sample_dropout_vs_oracle(T, B, feats, levels, shape, cin, fold):
    """batchsize > 1 with conv_dropout in bf16 (BASELINE.json configs[4]: `mixed bf16` fine-tuning takes batchsize / dropout like
    training()): one feature mask per SAMPLE (ext/neuron/models.py:320-324), the dropped-out tensors materialised in bf16
    (synthsr_scale_channels_bf16), the ELU backward with the per-sample factor (synthsr_elu_bwd_drop_bf16).  Against the oracle
    with the same [B, C] factors: (a) plain fp32 -- stated bf16 tolerances; (b) with the bf16 storage roundings restated -- tight."""
    torch = T
    from synthsr_amd.unet import unet
    from oracle import unet_ref as U
    rate = .3
    net = unet(nb_features=feats, input_shape=list(shape) + [cin], nb_levels=levels, conv_size=3, nb_labels=1, feat_mult=2,
               nb_conv_per_level=2, final_pred_activation='linear', batch_norm=-1, activation='elu', seed=3, dtype='bf16',
               fold_upsample=fold, conv_dropout=rate)
    g = torch.Generator().manual_seed(11)
    for nm, v in net.named_parameters():
        if nm.endswith('/gamma'):
            v.copy_(torch.rand(v.shape, generator=g) + .5)
        elif nm.endswith('/beta') or nm.endswith('/bias'):
            v.copy_(torch.randn(v.shape, generator=g) * .1)
    net.repack()
    net.set_batch(B)
    x = torch.rand(B, *shape, cin, generator=g)
    x[1] *= 1.7
    target = torch.rand(B, *shape, 1, generator=g)
    rng = np.random.default_rng(5)
    sc = {}
    for c in net.all_convs():
        keep = rng.random((B, c['cout'])) >= rate
        for b in range(B):  # every layer drops a feature of every sample, and not the same one
            keep[b, (3 * b + 1) % c['cout']] = False
            keep[b, (3 * b + 2) % c['cout']] = True
        sc[c['name']] = (keep / (1.0 - rate)).astype(np.float32)
    net.set_dropout_scales(sc)
    loss, pred = net.loss_l1(x.reshape(B * shape[0], shape[1], shape[2], cin).cuda(), target.reshape(-1).cuda(), want_pred=True)
    pred = pred.clone()
    net.backward()
    assert net.saved['encd'][0][0].dtype == torch.bfloat16 and net._drop_ps is not None
    for mode in ('fp32', 'bf16-storage'):
        P = {nm: v.detach().cpu().clone().requires_grad_(True) for nm, v in net.named_parameters()}
        stats = {}
        pr = U.unet_forward(x, P, net.prefix, levels, 2, training=True, collect=stats,
                            dropout={k: torch.from_numpy(v) for k, v in sc.items()},
                            quant=None if mode == 'fp32' else U.round_bf16)
        lr = U.l1_loss(pr, target)
        lr.backward()
        tight = mode != 'fp32'
        close(pred.view(B, *shape, 1), pr, 2.5e-2 if tight else 3e-2, mode + ' prediction')
        assert abs(loss.item() - lr.item()) < (3e-3 if tight else 1e-2) * abs(lr.item()), (mode, loss.item(), lr.item())
        for bn in net.bn_layers:      # statistics of the dropped-out tensors, over batch and voxels
            o, C = bn['soff'], bn['C']
            close(net.bn_batch[o:o + C], stats[bn['name']][0], 1.5e-2 if tight else 2e-2, mode + ' ' + bn['name'] + ' mean')
            close(net.bn_batch[o + C:o + 2 * C], stats[bn['name']][1], 1.5e-2 if tight else 2e-2, mode + ' ' + bn['name'] + ' var')
        rep = _grad_report(net, P)
        worst = sorted(((c, e, nm) for nm, (e, c, _) in rep.items()))[:4]
        for nm, (err, cos, kind) in rep.items():
            cmin, emax = ((0.985, 0.15) if fold else (0.997, 8e-2)) if tight else (0.9, 10.0)   # as test_unet_bf16_step_vs_oracle
            assert cos > cmin and err < emax, '%s: gradient of %s: err %.3e cos %.5f (worst %s)' % (mode, nm, err, cos, worst)
    # a feature dropped for EVERY sample has no gradient on the input-channel slice of the kernel that consumes it
    for grp in net.enc + net.dec:
        for k in range(1, len(grp['convs'])):
            dead = np.flatnonzero((sc[grp['convs'][k - 1]['name']] == 0).all(0))
            assert dead.size == 0 or net.view(grp['convs'][k]['w'], net.grads)[:, :, :, dead, :].abs().max().item() == 0.0


def test_bf16_training_reduces_loss(T):
    """a few bf16 steps of the full loop (generator -> bf16 U-Net -> Adam on fp32 master weights) on a fixed sample"""
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
               batch_norm=-1, seed=1, dtype='bf16')
    tr = Trainer(bg, net, lr=1e-3)
    inputs = next(bg.model_inputs_generator)
    draws = bg.labels_to_image_model.sample_draws()
    losses = [tr.step(inputs, draws).item() for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_training_entry_points_in_bf16(tmp_path):
    """training(dtype='bf16') and fine_tuning_with_adversary.training(dtype='bf16') (bf16 generator next to the fp32 critic:
    the "mixed bf16" of BASELINE.json configs[4]) end to end on a tiny problem: finite, decreasing loss, checkpoints written"""
    import os
    from synthsr_amd.training import training
    from synthsr_amd.fine_tuning_with_adversary import training as adv_training
    from synthsr_amd.nifti import write_nifti
    from synthsr_amd.synthetic import (GENERATION_LABELS, GENERATION_CLASSES, PRIOR_MEANS_T1_HR, PRIOR_STDS_T1_HR,
                                       synthetic_label_map)
    (tmp_path / 'labels').mkdir()
    (tmp_path / 'images').mkdir()
    rng = np.random.RandomState(0)
    lut = rng.uniform(30, 220, 64)
    for i in range(2):
        lab = synthetic_label_map((40, 36, 48), 10 + i)
        write_nifti(str(tmp_path / 'labels' / ('brain%d_labels.nii.gz' % i)), lab.astype(np.float32))
        write_nifti(str(tmp_path / 'images' / ('brain%d.nii.gz' % i)), (lut[lab % 64] + rng.randn(*lab.shape)).astype(np.float32))
    for nm, a in (('gl', GENERATION_LABELS), ('gc', GENERATION_CLASSES), ('pm', PRIOR_MEANS_T1_HR), ('ps', PRIOR_STDS_T1_HR)):
        np.save(tmp_path / (nm + '.npy'), a)
    model_dir = str(tmp_path / 'models')
    net = training(str(tmp_path / 'labels'), model_dir, str(tmp_path / 'pm.npy'), str(tmp_path / 'ps.npy'),
                   str(tmp_path / 'gl.npy'), path_generation_classes=str(tmp_path / 'gc.npy'), output_shape=32, n_levels=3,
                   unet_feat_count=24, nonlin_shape_factor=.125, bias_shape_factor=.125, steps_per_epoch=4, epochs=3,
                   verbose=False, lr=1e-3, dtype='bf16')
    assert net.bf16 and net.saved['enc'][0][0].dtype == __import__('torch').bfloat16
    log = [float(l.split(',')[1]) for l in open(os.path.join(model_dir, 'logs', 'loss.csv')).read().strip().split('\n')]
    assert len(log) == 3 and all(np.isfinite(log)) and log[-1] < log[0]
    # the TensorBoard event file KC.TensorBoard would have written (SynthSR/training.py:431): 'loss' at step = epoch index
    from synthsr_amd.tb_events import read_events
    ev = read_events(__import__('glob').glob(os.path.join(model_dir, 'logs', 'events.out.tfevents.*'))[0])
    assert ev[0]['file_version'] == 'brain.Event:2' and [e['step'] for e in ev[1:]] == [0, 1, 2]
    assert np.allclose([e['scalars'][0][1] for e in ev[1:]], log, rtol=1e-5, atol=1e-7) and ev[1]['scalars'][0][0] == 'loss'
    gen, critic = adv_training(str(tmp_path / 'labels'), str(tmp_path / 'images'), str(tmp_path / 'adv'), None, None,
                               str(tmp_path / 'gl.npy'), output_shape=32, n_levels=3, nonlin_shape_factor=.125,
                               bias_shape_factor=.125, epochs=1, steps_per_epoch=2, first_training_ratio=2,
                               training_ratio=1, lr_generator=1e-3, lr_discriminator=1e-3, verbose=False, dtype='bf16')
    assert gen.bf16 and gen.iterations == 2 and critic.iterations == 3
    assert np.all(np.isfinite(np.load(tmp_path / 'adv' / 'logs' / 'generator_loss.npy')))
    assert 'generator_1.h5' in os.listdir(tmp_path / 'adv')


@pytest.mark.parametrize('shape,n_filters,n_levels,masked', [((16, 16, 16), 8, 2, False), ((16, 16, 32), 32, 3, False),
                                                          ((16, 16, 16), 8, 2, True)])
def test_critic_bf16_vs_fp32_oracle(T, shape, n_filters, n_levels, masked):
    """WGAN-GP critic with the conv stack in bf16 (Critic3D(dtype='bf16')): D(x), the gradient norm, the loss, every
    parameter gradient (penalty term included) and the generator-side input gradient against the fp32 oracle under
    autograd with create_graph.  bf16 tolerances: scalars 2 %, gradients cosine >= 0.98 for kernels / 0.9 for biases (there is no pooling here, so no
    arg-max flips: LeakyReLU masks flip only where a pre-activation is within bf16 rounding of zero)"""
    torch = T
    from synthsr_amd.critic import Critic3D
    from oracle import unet_ref as U
    net = Critic3D(list(shape) + [1], n_filters=n_filters, n_levels=n_levels, seed=1, dtype='bf16')
    g = torch.Generator().manual_seed(7)
    for nm, _ in net.specs:
        v = net.view(nm)
        v.copy_((torch.randn(v.shape, generator=g) * (0.1 if nm.endswith('bias') else 1.0)).to(v.device) *
                (1.0 if nm.endswith('bias') else 3.0 * v.abs().max().item()))
    net.repack()
    real, fake = torch.rand(*shape, 1, generator=g), torch.rand(*shape, 1, generator=g)
    u = 0.3
    mask = dmask = None
    if masked:
        mask = (torch.rand(*shape, 1, generator=g) > 0.3).float()
        dmask = mask.cuda()
    loss, d_real, d_fake, norm = net.critic_loss_and_grads(real.cuda(), fake.cuda(), u, gp_weight=10.0, mask=dmask)
    P = {k: v.clone().requires_grad_(True) for k, v in net.state_dict().items()}
    ref, nref = U.critic_loss(real, fake, u, P, net.name, n_levels, 10.0, mask=mask)
    ref.backward()
    nref, ref = float(nref.detach()), float(ref.detach())
    assert abs(norm - nref) < 2e-2 * nref, (norm, nref)
    assert abs(loss - ref) < 5e-2 * max(1.0, abs(ref)), (loss, ref)    # ~ 2 x the error of the norm (penalty = 10 (1 - norm)^2)
    assert abs(nref - 1.0) > 0.05
    worst = []
    for nm, _ in net.specs:
        got, want = net.view(nm, net.grads).cpu().double().reshape(-1), P[nm].grad.double().reshape(-1)
        if float(want.norm()) < 1e-12:          # dense_1/bias: -1 (real) + 1 (fake) + 0 (penalty) = 0 exactly
            assert float(got.norm()) < 1e-6
            continue
        cos = float(torch.dot(got, want) / (got.norm() * want.norm()).clamp_min(1e-30))
        worst.append((cos, nm))
        # kernels >= 0.98 (measured >= 0.9888); biases are sums of cancelling signals over every voxel (the difference of the real and fake passes; measured 0.933 .. 0.9999): >= 0.9
        assert cos > (0.9 if nm.endswith('/bias') else 0.98), (nm, cos, sorted(worst)[:3])
    x = fake.clone().requires_grad_(True)
    d = U.critic_forward(x if mask is None else x * mask, {k: v.detach() for k, v in P.items()}, net.name, n_levels)
    gx, = torch.autograd.grad(d, x)
    got = net.input_gradient(fake.cuda(), dout=-0.01, mask=dmask).cpu().double().reshape(-1)
    want = (-0.01 * gx).double().reshape(-1)
    assert float(torch.dot(got, want) / (got.norm() * want.norm())) > 0.99
    before = net.forward(real.cuda()).item()
    net.adam_step(lr=1e-3)
    assert net.forward(real.cuda()).item() != before
