"""WGAN-GP critic (synthsr_amd/critic.py, csrc/critic.hip) against the oracle's torch restatement with autograd,
including the double backward of the gradient penalty (oracle/unet_ref.py:critic_loss).  SURVEY §8a U4 / §8f row 2."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def close(a, b, rel, name=''):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(b.abs().max().item(), 1e-30)
    err = (a - b).abs().max().item() / scale
    assert err < rel, '%s: max rel err %.3e (scale %.3e)' % (name, err, scale)


def test_critic_pieces():
    import torch
    from synthsr_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(6, 4, 8, 5, generator=g)
    y = ops.leaky_relu(x.cuda().clone())
    assert torch.equal(y.cpu(), torch.where(x > 0, x, 0.2 * x))
    dy = torch.randn(6, 4, 8, 5, generator=g)
    assert torch.equal(ops.leaky_relu_bwd(dy.cuda(), y).cpu(), dy * torch.where(x > 0, torch.ones(()), torch.full((), 0.2)))
    yb = ops.bias_leaky_relu(x.cuda().clone(), torch.arange(5.0).cuda() - 2)
    xb = x + (torch.arange(5.0) - 2)
    assert torch.equal(yb.cpu(), torch.where(xb > 0, xb, 0.2 * xb))
    cs = torch.zeros(5, device='cuda')
    ops.colsum(x.cuda(), cs)
    close(cs, x.reshape(-1, 5).sum(0), 1e-5, 'colsum')
    W, b, v = torch.randn(700, 96, generator=g), torch.randn(96, generator=g), torch.randn(700, generator=g)
    close(ops.dense_fwd(v.cuda(), W.cuda(), b.cuda()), v @ W + b, 1e-5, 'dense_fwd')
    dyv = torch.randn(96, generator=g)
    dx, dW = torch.empty(700, device='cuda'), torch.ones(700, 96, device='cuda')
    ops.dense_bwd(v.cuda(), W.cuda(), dyv.cuda(), dx=dx, dW=dW)
    close(dx, W @ dyv, 1e-5, 'dense dx')
    close(dW, 1 + v[:, None] * dyv[None, :], 1e-6, 'dense dW')
    s = torch.zeros(1, device='cuda')
    ops.sumsq(v.cuda(), s)
    assert abs(s.item() - float((v.double() ** 2).sum())) < 1e-3
    close(ops.axpby(v.cuda(), (2 * v).cuda(), 0.25, 0.5), 1.25 * v, 1e-6, 'axpby')


@pytest.mark.parametrize('shape,n_filters,n_levels,masked,K', [((16, 16, 16), 8, 2, False, 1), ((8, 16, 24), 32, 3, False, 1),
                                                            ((16, 16, 16), 8, 2, True, 1), ((16, 16, 16), 8, 2, True, 2),
                                                            ((8, 16, 24), 8, 2, False, 3)])
def test_critic_loss_and_gradients_vs_autograd(shape, n_filters, n_levels, masked, K):
    """-D(real) + D(fake) + 10 (1 - ||grad D(x_hat)||)^2 and its gradient w.r.t. every critic parameter (the penalty term
    through the masked forward pass) against autograd with create_graph; the input gradient used by the generator step"""
    import torch
    from synthsr_amd.critic import Critic3D
    from oracle import unet_ref as U
    net = Critic3D(list(shape) + [K], n_filters=n_filters, n_levels=n_levels, seed=1)   # K = output channels of the generator
    g = torch.Generator().manual_seed(7)
    for nm, _ in net.specs:                      # non-zero biases, larger weights (so that the penalty is active)
        v = net.view(nm)
        v.copy_((torch.randn(v.shape, generator=g) * (0.1 if nm.endswith('bias') else 1.0)).to(v.device) *
                (1.0 if nm.endswith('bias') else 3.0 * v.abs().max().item()))
    net.repack()
    real, fake = torch.rand(*shape, K, generator=g), torch.rand(*shape, K, generator=g)
    u = 0.3
    mask = None
    if masked:     # labels_to_mask: ConvertLabels(generation_labels, labels_to_mask)(segmentation) through the device LUT
        from synthsr_amd import ops
        seg = torch.randint(0, 6, shape, generator=g, dtype=torch.int32)
        lut = torch.tensor([0., 1., 1., 0., 1., 1.])
        mask = lut[seg.long()][..., None]
        dmask = ops.lut_gather(seg.cuda(), lut.cuda())[..., None].contiguous()
        assert torch.equal(dmask.cpu(), mask)
        dmask = dmask.expand(*shape, K).contiguous()      # the same mask on every channel (AdversarialTrainer._mask)
    loss, d_real, d_fake, norm = net.critic_loss_and_grads(real.cuda(), fake.cuda(), u, gp_weight=10.0,
                                                            mask=dmask if masked else None)
    P = {k: v.clone().requires_grad_(True) for k, v in net.state_dict().items()}
    ref, nref = U.critic_loss(real, fake, u, P, net.name, n_levels, 10.0, mask=mask)
    ref.backward()
    nref, ref = float(nref.detach()), float(ref.detach())
    assert abs(norm - nref) < 2e-4 * nref, (norm, nref)
    assert abs(loss - ref) < 2e-4 * max(1.0, abs(ref)), (loss, ref)
    assert abs(nref - 1.0) > 0.05            # the penalty contributes
    for nm, _ in net.specs:
        close(net.view(nm, net.grads), P[nm].grad, 2e-3, 'grad ' + nm)
    # generator side: -w * grad_x D(x)
    x = fake.clone().requires_grad_(True)
    d = U.critic_forward(x if mask is None else x * mask, {k: v.detach() for k, v in P.items()}, net.name, n_levels)
    gx, = torch.autograd.grad(d, x)
    close(net.input_gradient(fake.cuda(), dout=-0.01, mask=dmask if masked else None), -0.01 * gx, 2e-3, 'input gradient')
    # one Adam step moves the parameters and re-packs the conv weights (D changes)
    before = net.forward(real.cuda()).item()
    net.adam_step(lr=1e-3)
    assert net.forward(real.cuda()).item() != before


def test_adversarial_fine_tuning_end_to_end(tmp_path):
    """fine_tuning_with_adversary.training(): schedule (first_training_ratio critic updates, then training_ratio per
    generator update), randomise_res generation with real-image targets, both networks updated, logs and Keras-layout
    checkpoints written, a checkpoint resumes the generator; argument errors as in the reference"""
    import os
    import torch
    from synthsr_amd.fine_tuning_with_adversary import training, AdversarialTrainer
    from synthsr_amd.keras_h5 import load_keras_weights
    from synthsr_amd.nifti import write_nifti
    from synthsr_amd.synthetic import GENERATION_LABELS, synthetic_label_map
    shape = (40, 36, 48)
    ldir, idir = tmp_path / 'labels', tmp_path / 'images'
    ldir.mkdir(), idir.mkdir()
    rng = np.random.RandomState(0)
    lut = rng.uniform(30, 220, 64)
    for i in range(2):
        lab = synthetic_label_map(shape, 10 + i)
        write_nifti(str(ldir / ('brain%d_labels.nii.gz' % i)), lab.astype(np.float32))
        write_nifti(str(idir / ('brain%d.nii.gz' % i)), (lut[lab % 64] + rng.randn(*lab.shape)).astype(np.float32))
    np.save(tmp_path / 'gl.npy', GENERATION_LABELS)
    calls = []
    orig_c, orig_g = AdversarialTrainer.critic_step, AdversarialTrainer.generator_step
    AdversarialTrainer.critic_step = lambda self: (calls.append('c'), orig_c(self))[1]
    AdversarialTrainer.generator_step = lambda self: (calls.append('g'), orig_g(self))[1]
    try:
        gen, critic = training(str(ldir), str(idir), str(tmp_path / 'models'), None, None, str(tmp_path / 'gl.npy'),
                               output_shape=32, n_levels=3, nonlin_shape_factor=.125, bias_shape_factor=.125, epochs=2,
                               steps_per_epoch=2, first_training_ratio=3, training_ratio=2, lr_generator=1e-3,
                               lr_discriminator=1e-3, verbose=False)
    finally:
        AdversarialTrainer.critic_step, AdversarialTrainer.generator_step = orig_c, orig_g
    assert ''.join(calls) == 'cccg' + 'ccg' * 3                          # 100 / 10 in the reference's defaults
    assert gen.iterations == 4 and critic.iterations == 9
    mdir = str(tmp_path / 'models')
    d, g = np.load(os.path.join(mdir, 'logs', 'discriminator_loss.npy')), np.load(os.path.join(mdir, 'logs', 'generator_loss.npy'))
    assert d.shape == (2,) and g.shape == (2,) and np.isfinite(d).all() and np.isfinite(g).all()
    for f in ('generator_1.h5', 'generator_2.h5', 'generator_2.npz', 'discriminator_1.h5', 'discriminator_2.h5'):
        assert os.path.isfile(os.path.join(mdir, f)), f
    saved = load_keras_weights(os.path.join(mdir, 'discriminator_2.h5'))
    assert all(torch.equal(torch.from_numpy(saved[k]).reshape(v.shape), v) for k, v in critic.state_dict().items())
    assert critic.input_shape == [32, 32, 32, 1] and saved['discriminator_dense_0/kernel'].shape == (2 ** 3 * 256, 512)
    gen2, _ = training(str(ldir), str(idir), str(tmp_path / 'm2'), None, None, str(tmp_path / 'gl.npy'), output_shape=32,
                       n_levels=3, nonlin_shape_factor=.125, bias_shape_factor=.125, epochs=1, steps_per_epoch=1,
                       first_training_ratio=1, training_ratio=1, checkpoint_generator=os.path.join(mdir, 'generator_2.h5'),
                       lr_generator=0.0, verbose=False)
    a, b = gen.state_dict(), gen2.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a if 'moving' not in k)  # lr 0: the loaded weights, untouched
    # labels_to_mask: the critic only sees the voxels whose label maps to 1
    to_mask = (np.asarray(GENERATION_LABELS) > 0).astype(np.int32)
    gen3, critic3 = training(str(ldir), str(idir), str(tmp_path / 'm3'), None, None, str(tmp_path / 'gl.npy'),
                             output_shape=32, n_levels=3, nonlin_shape_factor=.125, bias_shape_factor=.125, epochs=1,
                             steps_per_epoch=1, first_training_ratio=2, training_ratio=1, labels_to_mask=to_mask,
                             verbose=False)
    assert critic3.mask_input and critic3.iterations == 2 and gen3.iterations == 1
    # two output channels (two synthetic input channels regressed onto themselves, no real images), masked critic
    gen4, critic4 = training(str(ldir), None, str(tmp_path / 'm4'), None, None, str(tmp_path / 'gl.npy'), input_channels=[True, True],
                             output_channel=[0, 1], output_shape=32, n_levels=3, nonlin_shape_factor=.125, bias_shape_factor=.125,
                             epochs=1, steps_per_epoch=1, first_training_ratio=2, training_ratio=1, labels_to_mask=to_mask,
                             verbose=False)
    assert critic4.input_shape == [32, 32, 32, 2] and gen4.nb_labels == 2 and critic4.iterations == 2 and gen4.iterations == 1
    g4 = np.load(os.path.join(str(tmp_path / 'm4'), 'logs', 'generator_loss.npy'))
    assert np.isfinite(g4).all()
    # batchsize 2: the generator runs on the stacked batch, the critic sample by sample (loss and gradients = batch means)
    gen5, critic5 = training(str(ldir), str(idir), str(tmp_path / 'm5'), None, None, str(tmp_path / 'gl.npy'), batchsize=2,
                             output_shape=32, n_levels=3, nonlin_shape_factor=.125, bias_shape_factor=.125, epochs=1,
                             steps_per_epoch=1, first_training_ratio=2, training_ratio=1, verbose=False)
    assert gen5.batch == 2 and critic5.iterations == 2 and gen5.iterations == 1
    assert np.isfinite(np.load(os.path.join(str(tmp_path / 'm5'), 'logs', 'generator_loss.npy'))).all()
    with pytest.raises(Exception, match='not both'):
        training(str(ldir), str(idir), mdir, None, None, str(tmp_path / 'gl.npy'), output_channel=0)
    with pytest.raises(Exception, match='output_channel or image_dir'):
        training(str(ldir), None, mdir, None, None, str(tmp_path / 'gl.npy'))


def test_critic_gradients_accumulate_over_the_samples_of_a_batch():
    """critic_loss_and_grads(accumulate=True) adds a sample's gradient to the buffer: two samples in a row = the sum of the two
    single-sample gradients (what AdversarialTrainer.critic_step divides by the batch size)"""
    import torch
    from synthsr_amd.critic import Critic3D
    shape = (16, 16, 16)
    net = Critic3D(list(shape) + [1], n_filters=8, n_levels=2, seed=3)
    g = torch.Generator().manual_seed(2)
    xs = [(torch.rand(*shape, 1, generator=g).cuda(), torch.rand(*shape, 1, generator=g).cuda()) for _ in range(2)]
    singles, losses = [], []
    for real, fake in xs:
        losses.append(net.critic_loss_and_grads(real, fake, 0.4, 10.0)[0])
        singles.append(net.grads.clone())
    l0 = net.critic_loss_and_grads(xs[0][0], xs[0][1], 0.4, 10.0)[0]
    l1 = net.critic_loss_and_grads(xs[1][0], xs[1][1], 0.4, 10.0, accumulate=True)[0]
    assert abs(l0 - losses[0]) < 1e-5 and abs(l1 - losses[1]) < 1e-5
    want = singles[0] + singles[1]
    assert float((net.grads - want).abs().max()) <= 1e-5 * float(want.abs().max())


def test_live_critic_and_unet_survive_a_change_of_conv_arithmetic():
    """packed weight layouts belong to the plan they were made for (arithmetic / plan options): repack() after
    ops.set_conv_arithmetic must re-plan and pack into fresh buffers (ops.conv_layout_epoch), for Critic3D as for UNet3D"""
    import torch
    from synthsr_amd import ops
    from synthsr_amd.critic import Critic3D
    from synthsr_amd.unet import unet
    shape = (32, 32, 32)
    net = Critic3D(list(shape) + [1], n_filters=24, n_levels=2, seed=3)
    gen = unet(24, list(shape) + [1], 2, 3, 1, feat_mult=2, nb_conv_per_level=2, batch_norm=-1, activation='elu',
               final_pred_activation='linear', seed=5)
    g = torch.Generator().manual_seed(2)
    real, fake = torch.rand(*shape, 1, generator=g).cuda(), torch.rand(*shape, 1, generator=g).cuda()
    x, t = torch.rand(*shape, 1, generator=g).cuda(), torch.rand(32 ** 3, generator=g).cuda()
    first = ops.conv_arithmetic()
    e0 = ops.conv_layout_epoch()
    l0 = net.critic_loss_and_grads(real, fake, 0.4, 10.0)[0]
    g0 = net.grads.clone()
    u0 = gen.loss_l1(x, t)[0].item()
    try:
        other = 'fp32_mfma' if first != 'fp32_mfma' else 'split'
        ops.set_conv_arithmetic(other)
        assert ops.conv_layout_epoch() != e0
        net.repack()
        gen.repack()
        l1 = net.critic_loss_and_grads(real, fake, 0.4, 10.0)[0]
        assert abs(l1 - l0) < 1e-4 * max(1.0, abs(l0))
        assert float((net.grads - g0).abs().max()) <= 2e-3 * float(g0.abs().max())
        assert abs(gen.loss_l1(x, t)[0].item() - u0) < 1e-5
    finally:
        ops.set_conv_arithmetic(first)
    e1 = ops.conv_layout_epoch()
    ops.set_conv_arithmetic(first)          # no change, no new epoch
    assert ops.conv_layout_epoch() == e1
    net.repack()
    gen.repack()
    l2 = net.critic_loss_and_grads(real, fake, 0.4, 10.0)[0]
    assert abs(l2 - l0) < 1e-5 * max(1.0, abs(l0))
    assert abs(gen.loss_l1(x, t)[0].item() - u0) < 1e-6


@pytest.mark.parametrize('lo_shape,cin,cout', [((4, 6, 8), 8, 16), ((8, 8, 8), 1, 32), ((6, 4, 10), 32, 32),
                                               ((4, 4, 8), 48, 24), ((8, 8, 16), 24, 48)])
def test_stride2_conv_on_the_parity_kernels(lo_shape, cin, cout):
    """stride-2 'same' Conv3D (TensorFlow padding (0, 1)) forward / data gradient / weight gradient / dbias through the
    parity kernels of the folded decoder conv (a second tap-to-slot table) against torch conv3d(stride=2) + autograd"""
    import torch
    import torch.nn.functional as F
    from synthsr_amd import ops
    g = torch.Generator().manual_seed(2)
    hi = tuple(2 * s for s in lo_shape)
    x = torch.randn(*hi, cin, generator=g, requires_grad=True)
    w = (torch.randn(3, 3, 3, cin, cout, generator=g) * 0.2).requires_grad_(True)
    xr = F.pad(x.permute(3, 0, 1, 2)[None], (0, 1, 0, 1, 0, 1))
    y_ref = F.conv3d(xr, w.permute(4, 3, 0, 1, 2), stride=2)[0].permute(1, 2, 3, 0)
    dy = torch.randn(*lo_shape, cout, generator=g)
    (y_ref * dy).sum().backward()
    wp = ops.pack_stride2_weights(w.detach().cuda(), lo_shape, 0)
    wpd = ops.pack_stride2_weights(w.detach().cuda(), lo_shape, 1)
    y = ops.conv3d_stride2(x.detach().cuda(), wp, cout)
    close(y, y_ref, 2e-5, 'forward')
    dx = ops.conv3d_stride2_dgrad(dy.cuda(), wpd, cin)
    close(dx, x.grad, 2e-5, 'data gradient')
    dw = torch.ones(3, 3, 3, cin, cout, device='cuda')
    db = torch.zeros(cout, device='cuda')
    dwc = torch.empty(8, 27, cout, cin, device='cuda')
    ops.conv3d_stride2_wgrad(x.detach().cuda(), dy.cuda(), dw, dwc, dbias=db)
    close(dw - 1, w.grad, 2e-5, 'weight gradient')
    close(db, dy.reshape(-1, cout).sum(0), 2e-5, 'dbias')
