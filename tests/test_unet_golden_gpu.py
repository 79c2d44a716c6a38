"""GPU parity of the HIP network path (through the C ABI) against the REFERENCE-DERIVED goldens of
tests/golden/unet_*.npz (the reference's own unet / metrics_model / add_seg_loss_to_model / make_discriminator
executed on the numpy Keras shim, see tests/golden/gen/make_unet_goldens.py) -- the same vectors that pin the oracle in
tests/test_unet_golden.py.  Tolerances: float32 conv stacks vs float64-evaluated goldens, 2e-4 of each tensor's range
(5e-4 for the 18-layer benchmark network); losses 2e-5 relative."""
import numpy as np
import pytest
from conftest import load_golden, regen_weights, golden_weights, tape_from_golden

pytestmark = pytest.mark.gpu

GEN = np.array([0, 14, 15, 16, 2, 3, 4, 5, 7, 8, 10, 11, 12, 13, 17, 18, 26, 28, 31], dtype=np.int32)


@pytest.fixture(scope='module')
def T():
    import torch
    assert torch.cuda.is_available()
    return torch


def close(a, b, rel=2e-4, name=''):
    import torch
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err < rel, '%s: max rel err %.3e (scale %.3e)' % (name, err, scale)


def _bn_stats(net, name):
    for bn in net.bn_layers:
        if bn['name'] == name:
            o, C = bn['soff'], bn['C']
            return net.bn_batch[o:o + C], net.bn_batch[o + C:o + 2 * C]
    raise KeyError(name)


@pytest.mark.parametrize('fold', [False, True])
def test_small_unet_vs_reference_wiring(T, fold):
    """3-level U-Net, anisotropic 16x24x8 two-channel input: training-phase prediction + every BatchNorm batch statistic,
    inference-phase prediction (moving statistics), softmax head"""
    torch = T
    from synthsr_amd.unet import unet
    g = load_golden('unet_wiring')
    x = torch.as_tensor(g['sm_train_x'][0]).cuda()
    kw = dict(nb_features=4, input_shape=[16, 24, 8, 2], nb_levels=3, conv_size=3, feat_mult=2, nb_conv_per_level=2,
              batch_norm=-1, activation='elu', fold_upsample=fold)
    net = unet(nb_labels=1, final_pred_activation='linear', **kw)
    net.load_state_dict(golden_weights(g, 'sm_train_w:'))
    zero = torch.zeros(16 * 24 * 8, device='cuda')
    _, pred = net.loss_l1(x, zero, want_pred=True)
    close(pred.view(16, 24, 8, 1), g['sm_train_pred'], name='training-phase prediction')
    for bn in net.bn_layers:
        m, v = _bn_stats(net, bn['name'])
        close(m, g['sm_train_bnmean:' + bn['name']], name=bn['name'] + ' mean')
        close(v, g['sm_train_bnvar:' + bn['name']], name=bn['name'] + ' var')
    net.load_state_dict(golden_weights(g, 'sm_infer_w:'))
    close(net.predict(x), g['sm_infer_pred'], name='inference-phase prediction')
    seg = unet(nb_labels=5, final_pred_activation='softmax', **kw)
    seg.load_state_dict(golden_weights(g, 'sm_softmax_w:'))
    close(seg.predict_probs(x).view(16, 24, 8, 5), g['sm_softmax_pred'], name='softmax posteriors')


@pytest.mark.parametrize('fold', [False, True])
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_dropout_vs_reference_wiring(T, fold, dtype):
    """conv_dropout: the reference's unet(conv_dropout=.4) on the shim in the learning phase, with the per-feature factors
    its Dropout layers drew handed to the device path (which never scales an activation: factors ride on the next conv's
    kernels and on the level's BatchNorm).  Prediction; the statistics the moving averages are fed = those of the dropped-out
    tensors the reference's BatchNormalization layers saw; inference ignores dropout."""
    torch = T
    from synthsr_amd.unet import unet
    g = load_golden('unet_dropout')
    x = torch.as_tensor(g['do_train_x'][0]).cuda()
    net = unet(nb_features=8, input_shape=[16, 8, 16, 2], nb_levels=3, conv_size=3, nb_labels=1, feat_mult=2,
               nb_conv_per_level=2, batch_norm=-1, activation='elu', final_pred_activation='linear', fold_upsample=fold,
               conv_dropout=.4, dtype=dtype)
    net.load_state_dict(golden_weights(g, 'do_train_w:'))
    sc = {}
    for l in range(3):
        for k in range(2):
            sc['unet_conv_downarm_%d_%d' % (l, k)] = g['do_train_scale:unet_dropout_downarm_%d_%d' % (l, k)]
    for lvl in range(2):
        for k in range(2):
            sc['unet_conv_uparm_%d_%d' % (3 + lvl, k)] = g['do_train_scale:unet_dropout_uparm_%d_%d' % (lvl, k)]
    net.set_dropout_scales(sc)
    zero = torch.zeros(16 * 8 * 16, device='cuda')
    _, pred = net.loss_l1(x, zero, want_pred=True)
    rel = 2e-4 if dtype == 'f32' else 4e-2
    close(pred.view(16, 8, 16, 1), g['do_train_pred'], rel, name='training-phase prediction with dropout')
    for bn in net.bn_layers:
        o, C = bn['soff'], bn['C']
        close(net.bn_true[o:o + C], g['do_train_bnmean:' + bn['name']], max(rel, 1e-3 if dtype == 'bf16' else rel),
              name=bn['name'] + ' mean of the dropped-out tensor')
        close(net.bn_true[o + C:o + 2 * C], g['do_train_bnvar:' + bn['name']], 5 * rel if dtype == 'bf16' else rel,
              name=bn['name'] + ' var of the dropped-out tensor')
    net.load_state_dict(golden_weights(g, 'do_infer_w:'))
    close(net.predict(x), g['do_infer_pred'], rel, name='inference-phase prediction')


@pytest.mark.parametrize('fold', [False, True])
def test_batch_of_volumes_vs_reference(T, fold):
    """batchsize 2 (UNet3D.set_batch: volumes stacked along the first axis) against the reference's unet run on a batch of
    two different volumes: prediction of both, BatchNorm statistics over batch and voxels"""
    torch = T
    from synthsr_amd.unet import unet
    g = load_golden('unet_batch')
    net = unet(nb_features=8, input_shape=[16, 8, 16, 2], nb_levels=3, conv_size=3, nb_labels=1, feat_mult=2,
               nb_conv_per_level=2, batch_norm=-1, activation='elu', final_pred_activation='linear', fold_upsample=fold)
    net.load_state_dict(golden_weights(g, 'b2_w:'))
    net.set_batch(2)
    x = torch.as_tensor(g['b2_x']).reshape(32, 8, 16, 2).cuda()
    _, pred = net.loss_l1(x, torch.zeros(2 * 16 * 8 * 16, device='cuda'), want_pred=True)
    close(pred.view(2, 16, 8, 16, 1), g['b2_pred'], name='prediction of the batch')
    for bn in net.bn_layers:
        m, v = _bn_stats(net, bn['name'])
        close(m, g['b2_bnmean:' + bn['name']], name=bn['name'] + ' mean')
        close(v, g['b2_bnvar:' + bn['name']], name=bn['name'] + ' var')


@pytest.mark.parametrize('fold', [False, True])
def test_batch_with_per_sample_dropout_vs_reference(T, fold):
    """batchsize 2 WITH conv_dropout: KL.Dropout(noise_shape=[None, 1, 1, 1, C]) draws one keep mask per sample and feature
    (ext/neuron/models.py:320-324); the reference's unet on the Keras shim, the factors [2, C] its Dropout layers drew handed
    to UNet3D.set_dropout_scales: prediction of both volumes, BatchNorm statistics of the dropped-out tensors"""
    torch = T
    from synthsr_amd.unet import unet
    from test_unet_golden import _dropout_scales
    g = load_golden('unet_batch_dropout')
    net = unet(nb_features=8, input_shape=[16, 8, 16, 2], nb_levels=3, conv_size=3, nb_labels=1, feat_mult=2,
               nb_conv_per_level=2, batch_norm=-1, activation='elu', final_pred_activation='linear', fold_upsample=fold,
               conv_dropout=.4)
    net.load_state_dict(golden_weights(g, 'b2d_w:'))
    net.set_batch(2)
    sc = {k: v.numpy() for k, v in _dropout_scales(g, 'b2d').items()}
    net.set_dropout_scales(sc)
    x = torch.as_tensor(g['b2d_x']).reshape(32, 8, 16, 2).cuda()
    _, pred = net.loss_l1(x, torch.zeros(2 * 16 * 8 * 16, device='cuda'), want_pred=True)
    close(pred.view(2, 16, 8, 16, 1), g['b2d_pred'], name='prediction of the batch')
    for bn in net.bn_layers:
        m, v = _bn_stats(net, bn['name'])
        close(m, g['b2d_bnmean:' + bn['name']], name=bn['name'] + ' mean')
        close(v, g['b2d_bnvar:' + bn['name']], name=bn['name'] + ' var')
    # ONE mask for both samples (what a batch of one draws) is a different network output
    net.set_dropout_scales({k: np.broadcast_to(v[:1], v.shape) for k, v in sc.items()})
    _, pred1 = net.loss_l1(x, torch.zeros(2 * 16 * 8 * 16, device='cuda'), want_pred=True)
    assert float((pred1.view(2, 16, 8, 16, 1).cpu() - torch.as_tensor(g['b2d_pred'])).abs().max()) > 1e-3


def test_training_graph_vs_reference(T):
    """the graph training() compiles (labels_to_image_model -> unet -> metrics_model, SynthSR/training.py:319-347) at 32^3
    with the benchmark network: HIP generator from the golden's labels + tape, HIP U-Net, fused head + L1 loss; plain and
    residual-channel + loss_cropping variants"""
    torch = T
    from synthsr_amd.labels_to_image_model import labels_to_image_model
    from synthsr_amd.unet import unet
    from test_generator_gpu import C2_KW
    g = load_golden('unet_training_graph')
    W = regen_weights(g['tg_w_names'], g['tg_w_shapes'], g['tg_w_seed'])
    for i, nm in enumerate(g['tg_w_names']):
        assert abs(W[str(nm)].astype(np.float64).sum() - g['tg_w_sum'][i]) < 1e-9 * max(1, g['tg_w_abs'][i]), nm
    net = unet(nb_features=24, input_shape=[32, 32, 32, 2], nb_levels=5, conv_size=3, nb_labels=1, feat_mult=2,
               nb_conv_per_level=2, batch_norm=-1, activation='elu', final_pred_activation='linear')
    assert net.n_params == int(g['tg_l1_n_trainable'])
    net.load_state_dict(W)
    for tag, residual, crop in (('tg_l1', None, None), ('tg_l1_res', 0, [24, 24, 16])):
        m = labels_to_image_model(labels_shape=[32, 32, 32], generation_labels=GEN, n_neutral_labels=len(GEN),
                                  aff=np.eye(4), output_shape=32, input_channels=[True], output_channel=[0], **C2_KW)
        draws = m.draws_from_tape(tape_from_golden(g, tag + '_tape'))
        image, target, seg = m.generate(g[tag + '_labels'][0, ..., 0], g[tag + '_means'][0], g[tag + '_stds'][0], draws)
        np.testing.assert_array_equal(seg.cpu().numpy(), g[tag + '_seg'][..., 0])
        np.testing.assert_allclose(image.cpu().numpy(), g[tag + '_image'], atol=2e-5)
        kw = {} if residual is None else dict(residual=image, res_stride=image.shape[-1], res_off=residual)
        loss, pred = net.loss(image, target.reshape(-1), 'l1', crop, want_pred=True, **kw)
        ref_pred = g[tag + '_unet_out'] + (0 if residual is None else g[tag + '_image'][..., residual:residual + 1])
        close(pred.view(32, 32, 32, 1), ref_pred, 5e-4, tag + ' prediction')
        assert abs(loss.item() - float(g[tag + '_loss'])) < 5e-5 * float(g[tag + '_loss']), (tag, loss.item())
        if tag == 'tg_l1':
            for bn in net.bn_layers:
                mm, vv = _bn_stats(net, bn['name'])
                close(mm, g['tg_bnmean:' + bn['name']], 5e-4, bn['name'] + ' mean')
                close(vv, g['tg_bnvar:' + bn['name']], 5e-4, bn['name'] + ' var')


def test_segmentation_loss_vs_reference(T):
    """metrics_model + add_seg_loss_to_model with the frozen softmax network (BatchNorm on moving statistics)"""
    torch = T
    from synthsr_amd.unet import unet
    from synthsr_amd.seg_loss import SegmentationRegulariser
    g = load_golden('unet_seg_loss')
    S = [16, 16, 16]
    kw = dict(nb_features=4, nb_levels=2, conv_size=3, feat_mult=2, nb_conv_per_level=2, batch_norm=-1, activation='elu')
    net = unet(input_shape=S + [2], nb_labels=1, final_pred_activation='linear', **kw)
    net.load_state_dict(golden_weights(g, 'sg_w:'))
    segnet = unet(input_shape=S + [1], nb_labels=len(g['sg_seg_labels']), final_pred_activation='softmax', **kw)
    segnet.load_state_dict(golden_weights(g, 'sg_segw:'))
    image = torch.as_tensor(g['sg_image']).cuda()
    target = torch.as_tensor(g['sg_target']).cuda().reshape(-1)
    seg = torch.as_tensor(g['sg_seg']).cuda()
    for i, tag in enumerate(str(c) for c in g['sg_cases']):
        m = None if np.isnan(g['sg_m'][i]) else float(g['sg_m'][i])
        M = None if np.isnan(g['sg_M'][i]) else float(g['sg_M'][i])
        crop = None if not g['sg_crop'][i].any() else [int(v) for v in g['sg_crop'][i]]
        for mode, key in (('inference', 'bninf'), ('batch', 'bnbatch')):   # the frozen network's BatchNorm: both Keras readings
            reg = SegmentationRegulariser(segnet, g['sg_gen_labels'], g['sg_seg_labels'], .25, m=m, M=M,
                                          fs_header=bool(g['sg_fs'][i]), frozen_bn=mode)
            loss, pred = net.loss(image, target, 'l1', crop, want_pred=True)
            close(pred.view(*S, 1), g[tag + '_bninf_pred'], name=tag + ' prediction')
            assert abs(loss.item() - float(g[tag + '_bninf_image_loss'])) < 1e-5
            dice = reg(pred, seg, net.dpred, crop)
            total = loss.item() + .25 * float(dice.item())
            assert abs(total - float(g[tag + '_%s_total' % key])) < 2e-5, (tag, mode, total, float(g[tag + '_%s_total' % key]))


@pytest.mark.parametrize('tag', ['cr_l4', 'cr_small', 'cr_mask'])
def test_critic_vs_reference(T, tag):
    """make_discriminator forward and the WGAN-GP critic loss through the HIP kernels"""
    torch = T
    from synthsr_amd.critic import Critic3D
    from test_unet_golden import _critic_params
    g = load_golden('unet_critic')
    P, n_levels = _critic_params(g, tag)
    real, fake = torch.as_tensor(g[tag + '_real']).cuda(), torch.as_tensor(g[tag + '_fake']).cuda()
    net = Critic3D(list(real.shape), n_filters=int(P['discriminator_conv_0/kernel'].shape[-1]), n_levels=n_levels)
    net.load_state_dict(P)
    mask = torch.as_tensor(g[tag + '_mask']).float().cuda().contiguous() if tag == 'cr_mask' else None  # int32 in the reference
    mk = 1.0 if mask is None else mask
    assert abs(net.forward((real * mk).contiguous()).item() - float(g[tag + '_d_real'])) < 5e-6
    assert abs(net.forward((fake * mk).contiguous()).item() - float(g[tag + '_d_fake'])) < 5e-6
    u = float(tape_from_golden(g, tag + '_tape')[0][1].reshape(-1)[0])
    loss, d_real, d_fake, norm = net.critic_loss_and_grads(real, fake, u, gp_weight=10.0, mask=mask)
    assert abs(norm - float(g[tag + '_grad_norm'])) < 2e-4 * float(g[tag + '_grad_norm']), (norm, float(g[tag + '_grad_norm']))
    assert abs(loss - float(g[tag + '_d_loss'])) < 1e-4 * abs(float(g[tag + '_d_loss']))
