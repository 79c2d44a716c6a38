"""Pins the NETWORK half of the oracle (oracle/unet_ref.py) and the product's layer tables against golden vectors
produced by executing the reference's own builders -- ext/neuron/models.py unet / conv_enc / conv_dec,
SynthSR/metrics_model.py metrics_model / add_seg_loss_to_model (+ DiceLoss), SynthSR/fine_tuning_with_adversary.py
make_discriminator / build_generator_loss / build_discriminator_loss -- on tests/golden/gen/keras_layers_shim.py
(tests/golden/gen/make_unet_goldens.py).  CPU only.

Pinned by these goldens: layer names, kernel shapes, parameter count, evaluation order, skip taps (pre-BatchNorm output
of conv_downarm_l_1), concatenation order [skip, up], BatchNorm placement, linear / softmax heads, residual channel
indexing, loss cropping, the Dice label-equivalency merge, critic layout, loss compositions.  NOT pinned (third-party
Keras / TensorFlow arithmetic restated from documentation in the shim): Conv3D / BatchNormalization / pooling / Dense
numerics, Adam, BatchNorm momentum.  Goldens are float64-evaluated and rounded to float32; the float32 oracle must
agree to 2e-5 of each tensor's range."""
import numpy as np
import pytest
import torch
from conftest import load_golden, regen_weights, golden_weights, tape_from_golden
from oracle import unet_ref as U


def tt(a):
    return torch.as_tensor(np.asarray(a))


def close(a, b, rel=2e-5, name=''):
    a = np.asarray(a.detach() if isinstance(a, torch.Tensor) else a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err < rel, '%s: max rel err %.3e (scale %.3e)' % (name, err, scale)


def test_layer_table_and_parameter_count_of_the_benchmark_network():
    """names / kernel shapes of training.py:330-341's network as the reference builder creates them == the product's
    parameter table; 13 240 489 trainable + 2 208 moving statistics = keras' count_params() 13 242 697"""
    from synthsr_amd.unet import UNet3D
    g = load_golden('unet_wiring')
    assert int(g['pc_n_trainable']) == 13240489 and int(g['pc_n_total']) == 13242697
    net = UNet3D(24, [16, 16, 16, 2], 5, 3, 1, feat_mult=2, nb_conv_per_level=2, batch_norm=-1, table_only=True)
    assert net.n_params == int(g['pc_n_trainable'])
    ref = {str(n): tuple(int(v) for v in s if v > 0) for n, s in zip(g['pc_names'], g['pc_shapes'])}
    ours = {nm: tuple(shp) for nm, shp, _ in net.specs}
    for nm, shp in ours.items():
        if nm.endswith('_likelihood/kernel'):               # keras stores the 1x1x1 head as [1,1,1,C,K]
            assert ref[nm] == (1, 1, 1) + shp
        else:
            assert ref[nm] == shp, nm
    moving = {n for n in ref if n.split('/')[-1].startswith('moving_')}
    assert set(ref) - moving == set(ours)
    assert {n.split('/')[0] for n in moving} == {b['name'] for b in net.bn_layers}
    # construction order of the conv / BN layers (the product stores BN as [beta | gamma]: compare layer order only)
    order_ref = [n.split('/')[0] for n in g['pc_names']]
    order_ours = [nm.split('/')[0] for nm, _, _ in net.specs]
    assert list(dict.fromkeys(order_ref)) == list(dict.fromkeys(order_ours))


@pytest.mark.parametrize('tag,training,softmax', [('sm_train', True, False), ('sm_infer', False, False),
                                                  ('sm_softmax', False, True)])
def test_oracle_unet_forward_vs_reference_wiring(tag, training, softmax):
    """3-level U-Net on an anisotropic two-channel volume: prediction, every BatchNorm layer's batch statistics and the
    evaluation order of the reference's builder"""
    g = load_golden('unet_wiring')
    W = golden_weights(g, tag + '_w:')
    P = {k: tt(v) for k, v in W.items()}
    P['unet_likelihood/kernel'] = P['unet_likelihood/kernel'].reshape(P['unet_likelihood/kernel'].shape[-2:])
    x = tt(g[tag + '_x'][0])
    stats = {}
    pred = U.unet_forward(x, P, 'unet', 3, 2, training=training, moving=P, collect=stats, softmax=softmax)
    close(pred, g[tag + '_pred'], name='prediction')
    if training:
        for name, (m, v) in stats.items():
            close(m, g['%s_bnmean:%s' % (tag, name)], name=name + ' mean')
            close(v, g['%s_bnvar:%s' % (tag, name)], name=name + ' var')
        assert len(stats) == 5
    order = [str(s) for s in g[tag + '_order']]
    assert order == ['unet_conv_downarm_0_0', 'unet_conv_downarm_0_1', 'unet_bn_down_0', 'unet_maxpool_0',
                     'unet_conv_downarm_1_0', 'unet_conv_downarm_1_1', 'unet_bn_down_1', 'unet_maxpool_1',
                     'unet_conv_downarm_2_0', 'unet_conv_downarm_2_1', 'unet_bn_down_2',
                     'unet_up_3', 'unet_merge_3', 'unet_conv_uparm_3_0', 'unet_conv_uparm_3_1', 'unet_bn_up_0',
                     'unet_up_4', 'unet_merge_4', 'unet_conv_uparm_4_0', 'unet_conv_uparm_4_1', 'unet_bn_up_1',
                     'unet_likelihood', 'unet_prediction']


def test_oracle_unet_intermediate_taps_vs_reference_wiring():
    """the skip tensors are the PRE-BatchNorm outputs of conv_downarm_l_1 and the merge is [skip, up] (models.py:431-434):
    recompute the reference's merge / conv outputs from its own stored activations with the oracle's primitives"""
    g = load_golden('unet_wiring')
    tag = 'sm_train'
    P = {k: tt(v) for k, v in golden_weights(g, tag + '_w:').items()}

    def act(n):
        return tt(g['%s_act:%s' % (tag, n)])
    merge = torch.cat([act('unet_conv_downarm_1_1'), U.upsample2(act('unet_bn_down_2'))], -1)
    close(merge, g[tag + '_act:unet_merge_3'], 1e-7, 'merge_3 = [skip pre-BN, up]')
    y, _, _ = U.batchnorm_train(act('unet_conv_downarm_0_1'), P['unet_bn_down_0/gamma'], P['unet_bn_down_0/beta'])
    close(y, g[tag + '_act:unet_bn_down_0'], name='bn_down_0')
    close(U.maxpool2(act('unet_bn_down_0')), g[tag + '_act:unet_maxpool_0'], 1e-7, 'maxpool_0')
    z = torch.nn.functional.elu(U.conv3d_same(act('unet_merge_4'), P['unet_conv_uparm_4_0/kernel'],
                                              P['unet_conv_uparm_4_0/bias']))
    close(z, g[tag + '_act:unet_conv_uparm_4_0'], name='conv_uparm_4_0')


def _training_graph_weights(g):
    W = regen_weights(g['tg_w_names'], g['tg_w_shapes'], g['tg_w_seed'])
    for i, nm in enumerate(g['tg_w_names']):
        a = W[str(nm)].astype(np.float64)
        assert abs(a.sum() - g['tg_w_sum'][i]) < 1e-9 * max(1, np.abs(a).sum()), nm
        assert abs(np.abs(a).sum() - g['tg_w_abs'][i]) < 1e-9 * max(1, np.abs(a).sum()), nm
    return W


def test_oracle_training_graph_vs_reference(gen_labels):
    """labels_to_image_model -> unet(input_model=...) -> metrics_model as training() wires them (training.py:319-347),
    32^3, the benchmark network: oracle generator + oracle U-Net + oracle loss from the same labels / tape / weights"""
    from oracle import generator_ref as R
    from test_oracle_golden import C2_KW
    g = load_golden('unet_training_graph')
    assert int(g['tg_l1_n_trainable']) == 13240489
    W = _training_graph_weights(g)
    P = {k: tt(v) for k, v in W.items()}
    P['unet_likelihood/kernel'] = P['unet_likelihood/kernel'].reshape(24, 1)
    for tag, residual, crop in (('tg_l1', None, None), ('tg_l1_res', 0, [24, 24, 16])):
        out = R.labels_to_image(g[tag + '_labels'][0, ..., 0], g[tag + '_means'][0], g[tag + '_stds'][0],
                                tape_from_golden(g, tag + '_tape'), gen_labels, len(gen_labels), output_shape=32,
                                input_channels=[True], output_channel=[0], **C2_KW)
        np.testing.assert_array_equal(out['seg'], g[tag + '_seg'][..., 0])
        np.testing.assert_allclose(out['image'], g[tag + '_image'], atol=5e-6)
        stats = {}
        image = tt(g[tag + '_image'])
        pred = U.unet_forward(image, P, 'unet', 5, 2, training=True, collect=stats)
        # 18 float32 conv layers with K up to 15552 and BatchNorm over as few as 8 voxels at the bottom level
        close(pred, g[tag + '_unet_out'], 2e-4, tag + ' prediction')
        res = None if residual is None else image[..., residual:residual + 1]
        loss = U.regression_loss(pred, tt(g[tag + '_target']), 'l1', loss_cropping=crop, residual=res)
        assert abs(float(loss) - float(g[tag + '_loss'])) < 2e-5 * float(g[tag + '_loss']), (tag, float(loss))
        if tag == 'tg_l1':
            for name, (m, v) in stats.items():
                close(m, g['tg_bnmean:' + name], 2e-4, name + ' mean')
                close(v, g['tg_bnvar:' + name], 2e-4, name + ' var')
            assert len(stats) == 9


def test_oracle_segmentation_loss_vs_reference():
    """metrics_model + add_seg_loss_to_model (metrics_model.py:136-215) with DiceLoss (layers.py:1264-1379): plain,
    clipped / normalised, FreeSurfer orientation, loss cropping.  'bninf' = the frozen network's BatchNorm uses its
    moving statistics; 'bnbatch' = batch statistics (Keras 2.3.1 under fit; the default of SegmentationRegulariser)"""
    g = load_golden('unet_seg_loss')
    P = {k: tt(v) for k, v in golden_weights(g, 'sg_w:').items()}
    P['unet_likelihood/kernel'] = P['unet_likelihood/kernel'].reshape(P['unet_likelihood/kernel'].shape[-2:])
    Ps = {k: tt(v) for k, v in golden_weights(g, 'sg_segw:').items()}
    Ps['unet_likelihood/kernel'] = Ps['unet_likelihood/kernel'].reshape(Ps['unet_likelihood/kernel'].shape[-2:])
    image, target, seg = tt(g['sg_image']), tt(g['sg_target']), tt(g['sg_seg'])
    for i, tag in enumerate(str(c) for c in g['sg_cases']):
        m = None if np.isnan(g['sg_m'][i]) else float(g['sg_m'][i])
        M = None if np.isnan(g['sg_M'][i]) else float(g['sg_M'][i])
        crop = None if not g['sg_crop'][i].any() else [int(v) for v in g['sg_crop'][i]]
        pred = U.unet_forward(image, P, 'unet', 2, 2, training=True)
        close(pred, g[tag + '_bninf_pred'], name=tag + ' prediction')
        image_loss = U.regression_loss(pred, target, 'l1', loss_cropping=crop)
        assert abs(float(image_loss) - float(g[tag + '_bninf_image_loss'])) < 2e-6
        dice = U.seg_regularisation(pred[..., 0], seg, Ps, 'unet', 2, 2, g['sg_gen_labels'], g['sg_seg_labels'],
                                    m=m, M=M, fs_header=bool(g['sg_fs'][i]), loss_cropping=crop)
        total = float(image_loss) + .25 * float(dice)
        assert abs(total - float(g[tag + '_bninf_total'])) < 5e-6, (tag, total, float(g[tag + '_bninf_total']))
        assert abs(float(g[tag + '_bnbatch_total']) - float(g[tag + '_bninf_total'])) > 1e-3   # the two modes differ
        # Keras' learning phase: the frozen network normalises with the statistics of its own activations
        dice_b = U.seg_regularisation(pred[..., 0], seg, Ps, 'unet', 2, 2, g['sg_gen_labels'], g['sg_seg_labels'],
                                      m=m, M=M, fs_header=bool(g['sg_fs'][i]), loss_cropping=crop, bn_batch_stats=True)
        total_b = float(image_loss) + .25 * float(dice_b)
        assert abs(total_b - float(g[tag + '_bnbatch_total'])) < 5e-6, (tag, total_b, float(g[tag + '_bnbatch_total']))


def _critic_params(g, tag):
    """keras auto-names (conv3d_1.., dense_1, dense_2: third-party naming) -> the product's / oracle's names"""
    W = golden_weights(g, tag + '_w:')
    convs = sorted({k.split('/')[0] for k in W if k.startswith('conv3d_')}, key=lambda s: int(s.split('_')[1]))
    dense = sorted({k.split('/')[0] for k in W if k.startswith('dense_')}, key=lambda s: int(s.split('_')[1]))
    P = {}
    for i, c in enumerate(convs):
        P['discriminator_conv_%d/kernel' % i] = tt(W[c + '/kernel'])
        P['discriminator_conv_%d/bias' % i] = tt(W[c + '/bias'])
    for i, d in enumerate(dense):
        P['discriminator_dense_%d/kernel' % i] = tt(W[d + '/kernel'])
        P['discriminator_dense_%d/bias' % i] = tt(W[d + '/bias'])
    return P, len(convs) // 2


def test_critic_layer_table_of_the_default_discriminator():
    g = load_golden('unet_critic')
    shapes = {str(n): tuple(int(v) for v in s if v > 0) for n, s in zip(g['cr_default_names'], g['cr_default_shapes'])}
    exp = []
    cin = 1
    for lvl in range(4):
        for _ in range(2):
            exp.append((3, 3, 3, cin, 32 * 2 ** lvl))
            cin = 32 * 2 ** lvl
    got = [shapes['conv3d_%d/kernel' % (i + 1)] for i in range(8)]
    assert got == exp
    assert shapes['dense_1/kernel'] == (256, 512) and shapes['dense_2/kernel'] == (512, 1)   # 16^3 -> 1^3 x 256
    assert int(g['cr_default_n_params']) == sum(int(np.prod(s)) for s in shapes.values())


@pytest.mark.parametrize('tag', ['cr_l4', 'cr_small', 'cr_mask'])
def test_oracle_critic_vs_reference(tag):
    """make_discriminator forward, the WGAN-GP critic loss (build_discriminator_loss with RandomWeightedAverage and
    Gradients) and build_generator_loss's composition"""
    g = load_golden('unet_critic')
    P, n_levels = _critic_params(g, tag)
    real, fake = tt(g[tag + '_real']), tt(g[tag + '_fake'])
    mask = tt(g[tag + '_mask']) if tag == 'cr_mask' else None
    mk = 1.0 if mask is None else mask
    d_real = U.critic_forward(real * mk, P, n_levels=n_levels)
    d_fake = U.critic_forward(fake * mk, P, n_levels=n_levels)
    for got, key in ((d_real, '_d_real'), (d_fake, '_d_fake')):
        assert abs(float(got) - float(g[tag + key])) < 2e-6, (tag, key, float(got), float(g[tag + key]))
    tape = tape_from_golden(g, tag + '_tape')
    assert len(tape) == 1 and tape[0][0] == 'u'
    u = float(tape[0][1].reshape(-1)[0])
    close(u * real + (1 - u) * fake, g[tag + '_averaged'], 1e-6, 'averaged samples')
    loss, norm = U.critic_loss(real, fake, u, P, n_levels=n_levels, gp_weight=10.0, mask=mask)
    assert abs(float(norm) - float(g[tag + '_grad_norm'])) < 2e-5 * float(g[tag + '_grad_norm'])
    assert abs(float(loss) - float(g[tag + '_d_loss'])) < 1e-5 * abs(float(g[tag + '_d_loss']))
    # generator loss: (1 - w_d) L1 + w_d mean(-D(G))  (no segmentation term), with and without loss cropping
    l1 = (real - fake).abs().mean()
    assert abs(float(.99 * l1 + .01 * -d_fake) - float(g[tag + '_g_loss'])) < 2e-6
    S = real.shape[:3]
    sl = tuple(slice(2, s - 2) for s in S)
    l1c = (real[sl] - fake[sl]).abs().mean()
    assert abs(float(.95 * l1c + .05 * -d_fake) - float(g[tag + '_g_loss_crop'])) < 2e-6
    if mask is not None:                                   # ConvertLabels(generation_labels, labels_to_mask)
        lut = dict(zip(g[tag + '_gen_labels'].tolist(), g[tag + '_labels_to_mask'].tolist()))
        exp = np.vectorize(lut.get)(g[tag + '_seg'])
        np.testing.assert_array_equal(g[tag + '_mask'][..., 0], exp)


def _dropout_scales(g, tag, nb_levels=3, nconv=2):
    """the factors the shim's Dropout layers drew, keyed by the conv layer each one follows: Keras names
    `unet_dropout_downarm_<l>_<k>` / `unet_dropout_uparm_<level>_<k>` (level counts the decoder stages from 0) ->
    `unet_conv_downarm_<l>_<k>` / `unet_conv_uparm_<nb_levels + level>_<k>` (ext/neuron/models.py:322, 449)"""
    sc = {}
    for l in range(nb_levels):
        for k in range(nconv):
            sc['unet_conv_downarm_%d_%d' % (l, k)] = tt(g['%s_scale:unet_dropout_downarm_%d_%d' % (tag, l, k)])
    for lvl in range(nb_levels - 1):
        for k in range(nconv):
            sc['unet_conv_uparm_%d_%d' % (nb_levels + lvl, k)] = tt(g['%s_scale:unet_dropout_uparm_%d_%d' % (tag, lvl, k)])
    return sc


def test_oracle_dropout_wiring_vs_reference():
    """conv_dropout: the reference's unet(conv_dropout=.4) run on the Keras shim in the learning phase, with the drawn
    per-feature factors recorded: the oracle given the same factors reproduces the prediction, the BatchNorm batch
    statistics (of the dropped-out tensors) and the concatenated tensors -- whose skip halves must be the conv layers' own
    outputs, NOT their dropped-out versions (ext/neuron/models.py:320-324, 431-434, 448-451).  Inference ignores dropout."""
    g = load_golden('unet_dropout')
    tag = 'do_train'
    P = {k: tt(v) for k, v in golden_weights(g, tag + '_w:').items()}
    P['unet_likelihood/kernel'] = P['unet_likelihood/kernel'].reshape(P['unet_likelihood/kernel'].shape[-2:])
    x = tt(g[tag + '_x'][0])
    sc = _dropout_scales(g, tag)
    assert all(np.all((v.numpy() == 0) | np.isclose(v.numpy(), 1 / .6, rtol=1e-6)) for v in sc.values())
    assert any((v == 0).any() for v in sc.values())
    stats = {}
    pred = U.unet_forward(x, P, 'unet', 3, 2, training=True, collect=stats, dropout=sc)
    close(pred, g[tag + '_pred'], name='prediction with dropout')
    for name, (m, v) in stats.items():
        close(m, g['%s_bnmean:%s' % (tag, name)], name=name + ' mean')
        close(v, g['%s_bnvar:%s' % (tag, name)], name=name + ' var')
    # the skip half of the first merge = ELU(conv_downarm_1_1(...)) un-dropped: it has no exact zeros although features
    # of that layer were dropped; the up-sampled half comes from a BatchNorm and has none either
    merge3 = g[tag + '_act:unet_merge_3']
    dropped = np.flatnonzero(sc['unet_conv_downarm_1_1'].numpy() == 0)
    assert dropped.size > 0 and np.abs(merge3[..., dropped]).max() > 0
    # without the factors the oracle must NOT match (the golden really exercises the dropout)
    plain = U.unet_forward(x, P, 'unet', 3, 2, training=True)
    assert float((plain - tt(g[tag + '_pred'])).abs().max()) > 1e-3
    # inference phase: dropout layers are the identity
    tag = 'do_infer'
    P = {k: tt(v) for k, v in golden_weights(g, tag + '_w:').items()}
    P['unet_likelihood/kernel'] = P['unet_likelihood/kernel'].reshape(P['unet_likelihood/kernel'].shape[-2:])
    pred = U.unet_forward(tt(g[tag + '_x'][0]), P, 'unet', 3, 2, training=False, moving=P)
    close(pred, g[tag + '_pred'], name='inference prediction')
    assert not any(k.startswith('do_infer_scale:') for k in g.files)


def test_oracle_batch_of_volumes_vs_reference():
    """batchsize 2: the reference's unet on the shim with a batch of two different volumes, learning phase -- prediction of
    both volumes and the BatchNorm statistics over batch and voxels"""
    g = load_golden('unet_batch')
    P = {k: tt(v) for k, v in golden_weights(g, 'b2_w:').items()}
    P['unet_likelihood/kernel'] = P['unet_likelihood/kernel'].reshape(P['unet_likelihood/kernel'].shape[-2:])
    stats = {}
    pred = U.unet_forward(tt(g['b2_x']), P, 'unet', 3, 2, training=True, collect=stats)
    close(pred, g['b2_pred'], name='prediction of the batch')
    for name, (m, v) in stats.items():
        close(m, g['b2_bnmean:' + name], name=name + ' mean')
        close(v, g['b2_bnvar:' + name], name=name + ' var')
    # per-volume statistics would differ: the second volume is 1.7x the first
    solo = U.unet_forward(tt(g['b2_x'][0]), P, 'unet', 3, 2, training=True)
    assert float((solo - tt(g['b2_pred'][0])).abs().max()) > 1e-3


def test_oracle_batch_with_per_sample_dropout_vs_reference():
    """batchsize 2 with conv_dropout: KL.Dropout(noise_shape=[None, 1, 1, 1, C]) draws one keep mask per SAMPLE and feature
    (ext/neuron/models.py:320-324); the reference's unet run on the Keras shim, its factors [2, C] handed to the oracle:
    prediction of both volumes and the BatchNorm statistics (over both samples of the dropped tensors)"""
    g = load_golden('unet_batch_dropout')
    P = {k: tt(v) for k, v in golden_weights(g, 'b2d_w:').items()}
    P['unet_likelihood/kernel'] = P['unet_likelihood/kernel'].reshape(P['unet_likelihood/kernel'].shape[-2:])
    sc = _dropout_scales(g, 'b2d')
    assert all(tuple(v.shape)[0] == 2 for v in sc.values())
    assert any(not np.array_equal(v[0].numpy(), v[1].numpy()) for v in sc.values())     # the two samples' masks differ
    stats = {}
    pred = U.unet_forward(tt(g['b2d_x']), P, 'unet', 3, 2, training=True, collect=stats, dropout=sc)
    close(pred, g['b2d_pred'], name='prediction of the batch')
    for name, (m, v) in stats.items():
        close(m, g['b2d_bnmean:' + name], name=name + ' mean')
        close(v, g['b2d_bnvar:' + name], name=name + ' var')
    same = {k: v[:1].expand(2, -1) for k, v in sc.items()}                             # ONE mask for both samples is different
    assert float((U.unet_forward(tt(g['b2d_x']), P, 'unet', 3, 2, training=True, dropout=same) - tt(g['b2d_pred'])).abs().max()) > 1e-3
