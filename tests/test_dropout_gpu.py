"""conv_dropout (ext/neuron/models.py:320-324, 448-451: feature-wise KL.Dropout after every conv + ELU): the device path
never scales an activation -- the per-channel factors ride on the next conv's input-channel weights and on the level's
BatchNorm (synthsr_amd/unet.py: _start_dropout / _dropout_bn) -- so the check is the whole network against the oracle,
which applies the same factors the way the reference graph does (multiplying the tensors), with autograd behind it."""
import numpy as np
import pytest

from conftest import single_shot_parity

pytestmark = pytest.mark.gpu


def _scales(net, rate, seed, force_drop=True):
    rng = np.random.default_rng(seed)
    out = {}
    for c in net.all_convs():
        keep = rng.random(c['cout']) >= rate
        if force_drop:
            keep[rng.integers(c['cout'])] = False  # every layer drops at least one feature
            keep[(np.flatnonzero(~keep)[0] + 1) % c['cout']] = True
        out[c['name']] = (keep / (1.0 - rate)).astype(np.float32)
    return out


@pytest.mark.parametrize('fold', [False, True])
@pytest.mark.parametrize('feats,levels,shape,cin,nconv,rate', [
    (24, 3, (16, 16, 32), 2, 2, .3), (8, 2, (8, 12, 16), 1, 2, .3), (8, 3, (16, 16, 16), 1, 1, .3), (8, 2, (8, 8, 16), 1, 3, .3),
    (8, 3, (16, 16, 16), 1, 1, 0.), (8, 2, (8, 8, 16), 2, 3, 0.)])   # rate 0: nb_conv_per_level 1 / 3 without dropout
def test_dropout_network_vs_oracle(feats, levels, shape, cin, nconv, rate, fold):
    import torch
    from synthsr_amd.unet import unet
    from oracle import unet_ref as U
    g = torch.Generator()
    tensors = {}

    def run():
        net = unet(nb_features=feats, input_shape=list(shape) + [cin], nb_levels=levels, conv_size=3, nb_labels=1,
                   feat_mult=2, nb_conv_per_level=nconv, final_pred_activation='linear', batch_norm=-1, activation='elu',
                   seed=3, fold_upsample=fold, conv_dropout=rate)
        g.manual_seed(11)
        for nm, v in net.named_parameters():
            if nm.endswith('/gamma'):
                v.copy_(torch.rand(v.shape, generator=g) + .5)
            elif nm.endswith('/beta') or nm.endswith('/bias'):
                v.copy_(torch.randn(v.shape, generator=g) * .1)
        net.repack()
        x = torch.rand(*shape, cin, generator=g)
        target = torch.rand(*shape, 1, generator=g)
        sc = _scales(net, rate, 5, force_drop=rate > 0)
        if rate > 0:
            net.set_dropout_scales(sc)
        loss, pred = net.loss_l1(x.cuda(), target.reshape(-1).cuda(), want_pred=True)
        net.test_loss, net.test_pred = loss.clone(), pred.clone()
        ready = []
        net.backward(on_grad_ready=ready.append)
        assert ready and ready == sorted(ready, reverse=True)      # the bucketing hook still fires, high offsets first
        tensors.update(x=x, target=target, sc=sc)
        return net

    def oracle(net, nudge):
        P = {nm: v.detach().cpu().clone().requires_grad_(True) for nm, v in net.named_parameters()}
        stats, pin = {}, []
        pr = U.unet_forward(tensors['x'], P, net.prefix, levels, nconv, training=True, collect=stats,
                            dropout={k: torch.from_numpy(v) for k, v in tensors['sc'].items()}, pool_inputs=pin,
                            pool_nudge=nudge)
        lr = U.l1_loss(pr, tensors['target'])
        lr.backward()
        tensors['stats'] = stats
        return (P, stats, pr.detach(), lr.detach()), pin

    def compare(net, ref):
        P, stats, pr, lr = ref
        err = (net.test_pred.view(*shape, 1).cpu() - pr).abs().max().item() / pr.abs().max().item()
        assert err < 5e-4, err
        assert abs(net.test_loss.item() - lr.item()) < 2e-5 * max(1.0, abs(lr.item()))
        # (every parameter gradient: the float64-anchored rule of conftest.single_shot_parity)

    net, _ = single_shot_parity(run, oracle, compare)
    x, sc = tensors['x'], tensors['sc']
    stats = tensors['stats']
    # a dropped feature has NO gradient on the input-channel slice of the kernel that consumes it
    for grp in net.enc + net.dec:
        for k in range(1, len(grp['convs'])):
            dead = np.flatnonzero(sc[grp['convs'][k - 1]['name']] == 0)
            assert dead.size == 0 or net.view(grp['convs'][k]['w'], net.grads)[:, :, :, dead, :].abs().max().item() == 0.0
    # the moving averages take the statistics of the dropped-out tensor (what the reference's BatchNormalization sees)
    mv0 = net.bn_moving.clone()
    net.update_moving_stats()
    for bn in net.bn_layers:
        o, C = bn['soff'], bn['C']
        m, v = stats[bn['name']]
        want_m = .99 * mv0[o:o + C].cpu() + .01 * m
        assert (net.bn_moving[o:o + C].cpu() - want_m).abs().max().item() < 1e-5
        corr = net.bn_corr[o + C:o + 2 * C].cpu()
        want_v = .99 * mv0[o + C:o + 2 * C].cpu() + .01 * v * corr
        assert (net.bn_moving[o + C:o + 2 * C].cpu() - want_v).abs().max().item() < 1e-5 * max(1.0, v.max().item())
    # the optimizer step packs the un-scaled kernels again: inference is dropout-free and matches the oracle's
    net.adam_step(lr=1e-3)
    assert not net._packed_scaled
    net.training = False
    out = net.predict(x.cuda())
    P2 = {nm: v.detach().cpu() for nm, v in net.named_parameters()}
    moving = {}
    for bn in net.bn_layers:
        o, C = bn['soff'], bn['C']
        moving[bn['name'] + '/moving_mean'] = net.bn_moving[o:o + C].cpu()
        moving[bn['name'] + '/moving_variance'] = net.bn_moving[o + C:o + 2 * C].cpu()
    ref = U.unet_forward(x, P2, net.prefix, levels, nconv, training=False, moving=moving)
    assert (out.cpu().view(*shape, 1) - ref).abs().max().item() < 5e-4 * max(1.0, ref.abs().max().item())


def test_dropout_draws_keep_rate_and_determinism():
    import torch
    from synthsr_amd.unet import unet
    kw = dict(nb_features=24, input_shape=[16, 16, 16, 1], nb_levels=3, conv_size=3, nb_labels=1, feat_mult=2,
              nb_conv_per_level=2, final_pred_activation='linear', batch_norm=-1, activation='elu', seed=4, conv_dropout=.25)
    a, b = unet(**kw), unet(**kw)
    x = torch.rand(16, 16, 16, 1).cuda()
    t = torch.rand(16 ** 3).cuda()
    kept, total = 0, 0
    for _ in range(6):
        la = a.loss_l1(x, t)[0]
        lb = b.loss_l1(x, t)[0]
        # same seed -> same masks (checked exactly below); the two networks drift apart by float-atomics noise only
        assert abs(la.item() - lb.item()) < 1e-3 * max(1.0, abs(la.item()))
        assert all(torch.equal(a._drop[k], b._drop[k]) for k in a._drop)
        for nm, s in a._drop.items():
            v = s.cpu().numpy()
            assert np.all((v == 0) | (np.abs(v - 1 / .75) < 1e-6))
            kept += int((s > 0).sum().item())
            total += s.numel()
        a.backward(); a.adam_step(); a.update_moving_stats()
        b.backward(); b.adam_step(); b.update_moving_stats()
    assert abs(kept / total - .75) < .05, kept / total
    assert torch.isfinite(a.params).all() and torch.isfinite(a.bn_moving).all()
    a.training = False
    assert torch.isfinite(a.predict(x)).all()


def test_dropout_masks_differ_between_forwards_of_one_iteration():
    """two training forwards WITHOUT an optimizer step in between (the critic updates of the adversarial schedule, repeated
    loss() calls) draw different masks, as Keras does on every forward; the FIRST forward of an iteration is a function of
    (seed, iteration) alone, so a network that skipped the extra forwards -- or was resumed -- sees the same mask there"""
    import torch
    from synthsr_amd.unet import unet
    kw = dict(nb_features=24, input_shape=[16, 16, 16, 1], nb_levels=3, conv_size=3, nb_labels=1, feat_mult=2,
              nb_conv_per_level=2, final_pred_activation='linear', batch_norm=-1, activation='elu', seed=9, conv_dropout=.5)
    a, b = unet(**kw), unet(**kw)
    x = torch.rand(16, 16, 16, 1).cuda()
    t = torch.rand(16 ** 3).cuda()
    a.loss_l1(x, t)
    first = {k: v.clone() for k, v in a._drop.items()}
    a.loss_l1(x, t)
    second = {k: v.clone() for k, v in a._drop.items()}
    assert any(not torch.equal(first[k], second[k]) for k in first)
    b.loss_l1(x, t)
    assert all(torch.equal(first[k], b._drop[k]) for k in first)       # first forward of iteration 0: same on both
    a.backward(); a.adam_step()
    b.backward(); b.adam_step()
    a.loss_l1(x, t)
    b.loss_l1(x, t)
    assert all(torch.equal(a._drop[k], b._drop[k]) for k in first)     # first forward of iteration 1: same again


def test_dropout_bf16_step_runs_and_matches_fp32_masks():
    """bf16 network: same mechanism (scaled kernels are re-packed to bf16); loose agreement with the fp32 network"""
    import torch
    from synthsr_amd.unet import unet
    kw = dict(nb_features=24, input_shape=[16, 16, 32, 2], nb_levels=3, conv_size=3, nb_labels=1, feat_mult=2,
              nb_conv_per_level=2, final_pred_activation='linear', batch_norm=-1, activation='elu', seed=4, conv_dropout=.2)
    f, h = unet(**kw), unet(dtype='bf16', **kw)
    sc = _scales(f, .2, 9)
    x = torch.rand(16, 16, 32, 2).cuda()
    t = torch.rand(16 * 16 * 32).cuda()
    f.set_dropout_scales(sc)
    h.set_dropout_scales(sc)
    lf, lh = f.loss_l1(x, t)[0], h.loss_l1(x, t)[0]
    assert abs(lf.item() - lh.item()) < 3e-2 * max(1.0, abs(lf.item()))
    f.backward()
    h.backward()
    gf, gh = f.grads.double(), h.grads.double()
    cos = (gf * gh).sum() / (gf.norm() * gh.norm())
    assert cos.item() > 0.97, cos.item()


def test_training_entry_point_accepts_dropout(tmp_path):
    import os
    from synthsr_amd.nifti import write_nifti
    from synthsr_amd.synthetic import (synthetic_label_map, GENERATION_LABELS, GENERATION_CLASSES, PRIOR_MEANS_T1_HR,
                                       PRIOR_STDS_T1_HR)
    from synthsr_amd.training import training
    d = tmp_path / 'labels'
    d.mkdir()
    for i in range(2):
        write_nifti(str(d / ('brain%d_labels.nii.gz' % i)), synthetic_label_map((40, 36, 48), 10 + i).astype(np.float32))
    for nm, v in (('gl', GENERATION_LABELS), ('gc', GENERATION_CLASSES), ('pm', PRIOR_MEANS_T1_HR), ('ps', PRIOR_STDS_T1_HR)):
        np.save(tmp_path / (nm + '.npy'), v)
    model_dir = str(tmp_path / 'models')
    net = training(str(d), model_dir, str(tmp_path / 'pm.npy'), str(tmp_path / 'ps.npy'), str(tmp_path / 'gl.npy'),
                   path_generation_classes=str(tmp_path / 'gc.npy'), output_shape=32, n_levels=3, unet_feat_count=24,
                   nonlin_shape_factor=.125, bias_shape_factor=.125, steps_per_epoch=3, epochs=1, dropout=.1, verbose=False)
    assert net.conv_dropout == .1 and net.iterations == 3 and net._drop is not None
    assert os.path.exists(os.path.join(model_dir, '001.npz'))
    log = open(os.path.join(model_dir, 'logs', 'loss.csv')).read().strip().split('\n')
    assert np.isfinite(float(log[0].split(',')[1]))
