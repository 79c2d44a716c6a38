"""Golden vectors for the NETWORK half of the path, produced by executing the reference's own builders
(read from /root/reference, never copied) on `keras_layers_shim`:

    ext/neuron/models.py:26-145,256-498        unet / conv_enc / conv_dec
    SynthSR/metrics_model.py:29-215            metrics_model, add_seg_loss_to_model  (+ ext/lab2im/layers.py DiceLoss)
    SynthSR/fine_tuning_with_adversary.py:482-642   make_discriminator, build_generator_loss,
                                                    build_discriminator_loss, RandomWeightedAverage, Gradients

Run in the development container only:   python tests/golden/gen/make_unet_goldens.py
Writes tests/golden/unet_*.npz (data only: inputs, weights, layer tables, activations, losses).

The few lines that glue the builders together below restate SynthSR/training.py:319-409 and
SynthSR/fine_tuning_with_adversary.py:320-433 (the argument lists of those calls); everything they call is the
reference's code.  See keras_layers_shim.py for what is pinned (wiring, names, shapes, loss composition) and what
remains third-party Keras arithmetic restated from its documentation.
"""
import os
import sys
import importlib.util
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.dirname(HERE)
REPO = os.path.dirname(os.path.dirname(OUT))
REF = '/root/reference'
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import tf_numpy_shim as shim  # noqa: E402
import keras_layers_shim as ks  # noqa: E402

FEED = []
ks.install(FEED, REF)

from ext.neuron import models as nrn_models  # noqa: E402
from ext.lab2im import layers as l2i_layers  # noqa: E402
from keras import models  # noqa: E402


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ref_l2i_model = _load('ref_l2i_model', 'SynthSR/labels_to_image_model.py')
ref_mm = _load('ref_metrics_model', 'SynthSR/metrics_model.py')
# fine_tuning_with_adversary.py uses a relative import of BrainGenerator at module level: give it a package stub
import types  # noqa: E402
_pkg = types.ModuleType('refpkg')
_pkg.__path__ = []
sys.modules['refpkg'] = _pkg
_bg = types.ModuleType('refpkg.brain_generator')
_bg.BrainGenerator = None
sys.modules['refpkg.brain_generator'] = _bg
_spec = importlib.util.spec_from_file_location('refpkg.fine_tuning_with_adversary',
                                               os.path.join(REF, 'SynthSR', 'fine_tuning_with_adversary.py'))
ref_adv = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ref_adv)

from synthsr_amd.nifti import read_nifti  # noqa: E402

GEN_LABELS = np.load(os.path.join(REF, 'data', 'labels_classes_priors', 'generation_labels.npy'))
GEN_CLASSES = np.load(os.path.join(REF, 'data', 'labels_classes_priors', 'generation_classes.npy'))


def tape_to_dict(tape, prefix='tape'):
    d = {'%s_kinds' % prefix: np.array([k for k, _ in tape.entries])}
    for i, (_, a) in enumerate(tape.entries):
        d['%s_%02d' % (prefix, i)] = a
    return d


def load_label_crop(which=1, origin=(60, 70, 60), shape=(32, 32, 32)):
    lab, aff, _ = read_nifti(os.path.join(REF, 'data', 'labels', 'brain%d_labels.nii.gz' % which))
    lab = np.round(lab).astype(np.int32)
    sl = tuple(slice(o, o + s) for o, s in zip(origin, shape))
    return np.ascontiguousarray(lab[sl])


def class_stats(rng, nm='t1_hr'):
    pm = np.load(os.path.join(REF, 'data', 'labels_classes_priors', 'prior_means_%s.npy' % nm))
    ps = np.load(os.path.join(REF, 'data', 'labels_classes_priors', 'prior_stds_%s.npy' % nm))
    m = np.clip(rng.normal(pm[0], pm[1]), 0, None)[GEN_CLASSES]
    s = np.clip(rng.normal(ps[0], ps[1]), 0, None)[GEN_CLASSES]
    return m[None, :, None].astype(np.float32), s[None, :, None].astype(np.float32)

t = shim.t


def layer_table():
    """name, class, kernel shape of every layer with weights, in construction order"""
    names, shapes = [], []
    for k, v in ks.PARAMS.items():
        names.append(k)
        shapes.append(list(v.shape) + [0] * (5 - v.ndim))
    return np.array(names), np.array(shapes, dtype=np.int64)


def n_trainable():
    return int(sum(v.size for k, v in ks.PARAMS.items() if not k.split('/')[-1].startswith('moving_')))


def params_dict(prefix='w:'):
    return {prefix + k: v for k, v in ks.PARAMS.items()}


def call_order():
    """layer names in the order the graph evaluated them (named layers only)"""
    return np.array([getattr(l, 'name', None) or '' for l, _, _ in ks.GRAPH])


TRAIN_DEFAULTS = dict(atlas_res=[1., 1., 1.], target_res=None, output_div_by_n=32, padding_margin=None,
                      flipping=True, scaling_bounds=.15, rotation_bounds=15, shearing_bounds=.02,
                      translation_bounds=5, nonlin_std=4., nonlin_shape_factor=.03125 * 4,
                      simulate_registration_error=True, randomise_res=False, data_res=None, thickness=None,
                      downsample=True, build_reliability_maps=True, blur_range=1.15, bias_field_std=.3,
                      bias_shape_factor=.03125 * 4)


# ------------------------------------------------------------------------------------------------ 1. parameter count
def golden_param_count():
    """the network of SynthSR/training.py:330-341 (24 features, x2 per level, 5 levels, 2 convs, BN, ELU, linear head)
    on a 16^3 x 2 input: layer names / kernel shapes and the number of trainable parameters (spatial size does not
    enter) -- must equal SURVEY's 13 242 697."""
    ks.reset(seed=1)
    FEED[:] = [('unet_input', np.zeros((1, 16, 16, 16, 2), np.float32))]
    model = nrn_models.unet(nb_features=24, input_shape=[16, 16, 16, 2], nb_levels=5, conv_size=3, nb_labels=1,
                            feat_mult=2, nb_conv_per_level=2, conv_dropout=0, final_pred_activation='linear',
                            batch_norm=-1, activation='elu', input_model=None)
    names, shapes = layer_table()
    n = n_trainable()
    print('param count', n, 'layers', len(names), 'out', np.asarray(model.output).shape)
    n_total = int(sum(v.size for v in ks.PARAMS.values()))       # keras' count_params(): moving statistics included
    assert n == 13240489 and n_total == 13242697, (n, n_total)   # SURVEY U1 quotes the total
    return dict(pc_names=names, pc_shapes=shapes, pc_n_trainable=np.int64(n), pc_n_total=np.int64(n_total),
                pc_call_order=call_order())


# ------------------------------------------------------------------------------------------------ 2. small net, all taps
def golden_small_unet(out):
    """3-level, 4-feature U-Net on an anisotropic 16x24x8 two-channel input: EVERY layer output is stored, in training
    phase (batch statistics) and in inference phase (moving statistics), for the linear head and for the softmax head"""
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 1, (1, 16, 24, 8, 2)).astype(np.float32)
    for tag, phase, nb_labels, final in (('sm_train', 1, 1, 'linear'), ('sm_infer', 0, 1, 'linear'),
                                         ('sm_softmax', 0, 5, 'softmax')):
        ks.reset(seed=2, learning_phase=phase)
        FEED[:] = [('unet_input', x)]
        model = nrn_models.unet(nb_features=4, input_shape=[16, 24, 8, 2], nb_levels=3, conv_size=3,
                                nb_labels=nb_labels, feat_mult=2, nb_conv_per_level=2, conv_dropout=0,
                                final_pred_activation=final, batch_norm=-1, activation='elu', input_model=None)
        out[tag + '_x'] = x
        out.update(params_dict(tag + '_w:'))
        out[tag + '_order'] = call_order()
        for layer, _, o in ks.GRAPH:
            out['%s_act:%s' % (tag, layer.name)] = np.asarray(o)[0]
        out[tag + '_pred'] = np.asarray(model.output)[0]
        for k, (m, v) in ks.STATE['bn_batch'].items():
            out['%s_bnmean:%s' % (tag, k)] = m
            out['%s_bnvar:%s' % (tag, k)] = v
        print(tag, 'pred', out[tag + '_pred'].shape, 'layers', len(ks.GRAPH))


# ------------------------------------------------------------------------------------------------ 3. whole training graph
def golden_training_graph(out):
    """labels_to_image_model -> unet(input_model=...) -> Model -> metrics_model, i.e. the graph training() compiles
    (SynthSR/training.py:319-347), at 32^3 with the benchmark network (24..384 features, 5 levels, Cin = 2)."""
    rng = np.random.default_rng(43)
    lab = load_label_crop(1, (58, 78, 62), (32, 32, 32))[None, ..., None]
    means, stds = class_stats(rng)
    for tag, metric, residual, crop in (('tg_l1', 'l1', None, None), ('tg_l1_res', 'l1', [0], [24, 24, 16])):
        ks.reset(seed=3)
        tape = shim.Tape(seed=171)
        shim.set_tape(tape)
        FEED[:] = [('labels_input', lab), ('means_input', means), ('std_devs_input', stds)]
        l2i = ref_l2i_model.labels_to_image_model(labels_shape=[32, 32, 32], input_channels=[True], output_channel=[0],
                                                  generation_labels=GEN_LABELS, n_neutral_labels=len(GEN_LABELS),
                                                  output_shape=32, aff=np.eye(4), **TRAIN_DEFAULTS)
        unet_input_shape = l2i.output[0].get_shape().as_list()[1:]
        model = nrn_models.unet(nb_features=24, input_shape=unet_input_shape, nb_levels=5, conv_size=3, nb_labels=1,
                                feat_mult=2, nb_conv_per_level=2, conv_dropout=0, final_pred_activation='linear',
                                batch_norm=-1, activation='elu', input_model=l2i)
        model = models.Model(model.inputs, model.output)
        wres = residual
        if wres is not None:                       # training.py:270-271 (build_reliability_maps doubles the LIST, F11)
            wres = 2 * wres
        pred = np.asarray(model.output)
        loss_model = ref_mm.metrics_model(input_model=model, loss_cropping=crop, metrics=metric,
                                          work_with_residual_channel=wres)
        out.update({tag + '_labels': lab, tag + '_means': means, tag + '_stds': stds,
                    tag + '_image': np.asarray(l2i.outputs[0])[0], tag + '_target': np.asarray(l2i.outputs[1])[0],
                    tag + '_seg': np.asarray(model.get_layer('segmentation_target').output)[0],
                    tag + '_unet_out': pred[0], tag + '_loss': np.float32(np.asarray(loss_model.outputs[0])),
                    tag + '_n_trainable': np.int64(n_trainable())})
        out.update(tape_to_dict(tape, tag + '_tape'))
        if tag == 'tg_l1':
            # 13.2 M weights are not stored: tests regenerate them with `regen_weights` (same recipe as
            # keras_layers_shim._weight, seed 3, in the order of tg_w_names) and check these checksums
            names, shapes = layer_table()
            out.update(tg_w_names=names, tg_w_shapes=shapes, tg_w_seed=np.int64(3),
                       tg_w_sum=np.array([np.float64(v.astype(np.float64).sum()) for v in ks.PARAMS.values()]),
                       tg_w_abs=np.array([np.float64(np.abs(v.astype(np.float64)).sum()) for v in ks.PARAMS.values()]))
            for k, (m, v) in ks.STATE['bn_batch'].items():
                out['tg_bnmean:' + k], out['tg_bnvar:' + k] = m, v
        print(tag, 'loss', out[tag + '_loss'], 'unet_input_shape', unet_input_shape, 'params', n_trainable())


# ------------------------------------------------------------------------------------------------ 4. segmentation loss
def golden_seg_loss(out):
    """metrics_model + add_seg_loss_to_model (SynthSR/training.py:344-409): small trained net + frozen segmentation net
    (softmax head) on a 16^3 volume; with / without clipping (m, M), FreeSurfer orientation, loss cropping.  The frozen
    network's BatchNorm is recorded both ways: 'bninf' = moving statistics, 'bnbatch' = batch statistics (Keras 2.3.1
    does not look at `trainable` in BatchNormalization.call as far as its documentation goes; unpinned)."""
    rng = np.random.default_rng(7)
    S = [16, 16, 16]
    image = rng.uniform(0, 1, [1] + S + [2]).astype(np.float32)
    target = rng.uniform(0, 1, [1] + S + [1]).astype(np.float32)
    gen_labels = np.array([0, 2, 3, 4, 41, 42, 43])
    seg_labels = np.array([0, 2, 3, 41, 42, 2, 77])          # label 2 appears twice (merged), 4 and 43 have no equivalent
    seg = rng.integers(0, len(gen_labels), [1] + S).astype(np.int32)       # NB indices, see metrics_model.py:191
    out.update(sg_image=image[0], sg_target=target[0], sg_seg=seg[0], sg_gen_labels=gen_labels, sg_seg_labels=seg_labels)
    cases = (('sg_plain', None, None, False, None), ('sg_clip', .2, .7, False, None),
             ('sg_fs', .1, .9, True, None), ('sg_crop', None, None, False, [8, 12, 16]))
    for tag, m, M, fs, crop in cases:
        for bn_mode in ('bninf', 'bnbatch'):
            ks.reset(seed=4)
            ks.STATE['frozen_bn_inference'] = bn_mode == 'bninf'
            # stand-in for the generator model: taps with the names metrics_model looks up
            FEED[:] = [('gen_image', image), ('gen_target', target), ('gen_seg', seg[..., None])]
            import keras.layers as KL
            im_in = KL.Input(shape=S + [2], name='gen_image')
            tg_in = KL.Input(shape=S + [1], name='gen_target')
            sg_in = KL.Input(shape=S + [1], name='gen_seg', dtype='int32')
            seg_t = KL.Lambda(lambda x: x + 0, name='segmentation_target')(sg_in)
            # same dependency tricks as labels_to_image_model.py:258-262, so that the taps are ancestors of the output
            tg = KL.Lambda(lambda x: x[0] + 0., name='regression_target')([tg_in, seg_t])
            im = KL.Lambda(lambda x: x[0] + 0., name='image_out')([im_in, tg])
            gen = models.Model(inputs=[im_in, tg_in, sg_in], outputs=[im, tg])
            model = nrn_models.unet(nb_features=4, input_shape=S + [2], nb_levels=2, conv_size=3, nb_labels=1,
                                    feat_mult=2, nb_conv_per_level=2, conv_dropout=0, final_pred_activation='linear',
                                    batch_norm=-1, activation='elu', input_model=gen)
            model = models.Model(model.inputs, model.output)
            model = ref_mm.metrics_model(input_model=model, loss_cropping=crop, metrics='l1',
                                         work_with_residual_channel=None)
            image_loss = np.float32(np.asarray(model.outputs[0]))
            gen_params = dict(ks.PARAMS)         # the segmentation net reuses the layer names ('unet_...'), as in training()
            ks.PARAMS.clear()
            FEED[:] = [('unet_input', np.zeros([1] + S + [1], np.float32))]
            seg_model = nrn_models.unet(nb_features=4, input_shape=S + [1], nb_levels=2, conv_size=3,
                                        nb_labels=len(seg_labels), feat_mult=2, nb_conv_per_level=2, conv_dropout=0,
                                        final_pred_activation='softmax', batch_norm=-1, activation='elu',
                                        input_model=None)
            seg_model.trainable = False
            for layer in seg_model.layers:
                layer.trainable = False
            total = ref_mm.add_seg_loss_to_model(input_model=model, seg_model=seg_model, generation_labels=gen_labels,
                                                 segmentation_label_equivalency=seg_labels, rel_weight=.25,
                                                 loss_cropping=crop, m=m, M=M, fs_header=fs)
            total_loss = np.float32(np.asarray(total.outputs[0]))
            out['%s_%s_total' % (tag, bn_mode)] = total_loss
            out['%s_%s_image_loss' % (tag, bn_mode)] = image_loss
            out['%s_%s_pred' % (tag, bn_mode)] = np.asarray(model.get_layer('predicted_image').output)[0]
            if tag == 'sg_plain' and bn_mode == 'bninf':
                out.update({'sg_w:' + k: v for k, v in gen_params.items()})
                out.update(params_dict('sg_segw:'))
            print(tag, bn_mode, 'image loss', image_loss, 'total', total_loss, 'dice', (total_loss - image_loss) / .25)
    out['sg_cases'] = np.array([c[0] for c in cases])
    out['sg_m'] = np.array([np.nan if c[1] is None else c[1] for c in cases])
    out['sg_M'] = np.array([np.nan if c[2] is None else c[2] for c in cases])
    out['sg_fs'] = np.array([c[3] for c in cases])
    out['sg_crop'] = np.array([[0, 0, 0] if c[4] is None else c[4] for c in cases])


# ------------------------------------------------------------------------------------------------ 5. critic
def golden_critic(out):
    """make_discriminator (+ mask input), build_discriminator_loss with the gradient penalty (K.gradients through the
    recorded critic) and build_generator_loss, 16^3 one-channel volumes, n_levels 4 / n_filters 32 (the defaults) and
    a 2-level 8-filter critic on an anisotropic 8x12x16 volume."""
    rng = np.random.default_rng(9)
    # the default critic (n_filters 32, n_levels 4): layer table and parameter count only (3.6 M weights not stored)
    ks.reset(seed=6)
    FEED[:] = [('input_discriminator', np.zeros([1, 16, 16, 16, 1], np.float32))]
    ref_adv.make_discriminator([16, 16, 16, 1])
    out['cr_default_names'], out['cr_default_shapes'] = layer_table()
    out['cr_default_n_params'] = np.int64(n_trainable())
    for tag, S, kw, use_mask in (('cr_l4', [16, 16, 16], dict(n_filters=8), False),
                                 ('cr_small', [8, 12, 16], dict(n_filters=8, n_levels=2), False),
                                 ('cr_mask', [8, 12, 16], dict(n_filters=8, n_levels=2), True)):
        ks.reset(seed=6)
        tape = shim.Tape(seed=181)
        shim.set_tape(tape)
        real = rng.uniform(0, 1, [1] + S + [1]).astype(np.float32)
        fake = rng.uniform(0, 1, [1] + S + [1]).astype(np.float32)
        gen_labels = np.array([0, 2, 3, 4, 41])
        labels_to_mask = np.array([0, 1, 1, 0, 1])
        seg = gen_labels[rng.integers(0, len(gen_labels), [1] + S)].astype(np.int32)
        FEED[:] = [('input_discriminator', np.zeros([1] + S + [1], np.float32))]
        if use_mask:
            FEED.append(('input_mask', np.zeros([1] + S + [1], np.float32)))
        disc = ref_adv.make_discriminator(S + [1], mask_input=use_mask, **kw)
        names, shapes = layer_table()
        real_t, fake_t = t(real), t(fake)
        averaged = ref_adv.RandomWeightedAverage()([real_t, fake_t])
        if use_mask:
            mask = l2i_layers.ConvertLabels(gen_labels, labels_to_mask, name='mask')(t(seg[..., None]))
            d_real, d_fake, d_av = disc([real_t, mask]), disc([fake_t, mask]), disc([averaged, mask])
            out[tag + '_mask'] = np.asarray(mask)[0]
        else:
            d_real, d_fake, d_av = disc(real_t), disc(fake_t), disc(averaged)
        d_loss = ref_adv.build_discriminator_loss(d_real, d_fake, d_av, averaged, 10, 3)
        g_loss = ref_adv.build_generator_loss(real_t, None, fake_t, d_fake, None, gen_labels, None, None, False, .25, .01)
        g_loss_crop = ref_adv.build_generator_loss(real_t, None, fake_t, d_fake, None, gen_labels, None,
                                                   [s - 4 for s in S], False, .25, .05)
        g_av = ks.gradients(d_av, averaged)[0]
        out[tag + '_grad_norm'] = np.float32(np.sqrt(np.sum(np.square(np.asarray(g_av, dtype=np.float64)))))
        out[tag + '_grad_av'] = np.asarray(g_av)[0]
        out.update({tag + '_real': real[0], tag + '_fake': fake[0], tag + '_seg': seg[0],
                    tag + '_averaged': np.asarray(averaged)[0], tag + '_names': names, tag + '_shapes': shapes,
                    tag + '_d_real': np.float32(np.asarray(d_real).reshape(())),
                    tag + '_d_fake': np.float32(np.asarray(d_fake).reshape(())),
                    tag + '_d_av': np.float32(np.asarray(d_av).reshape(())),
                    tag + '_d_loss': np.float32(np.asarray(d_loss)), tag + '_g_loss': np.float32(np.asarray(g_loss)),
                    tag + '_g_loss_crop': np.float32(np.asarray(g_loss_crop)),
                    tag + '_n_params': np.int64(n_trainable()), tag + '_gen_labels': gen_labels,
                    tag + '_labels_to_mask': labels_to_mask})
        out.update(params_dict(tag + '_w:'))
        out.update(tape_to_dict(tape, tag + '_tape'))
        print(tag, 'D', out[tag + '_d_real'], out[tag + '_d_fake'], out[tag + '_d_av'], 'd_loss', out[tag + '_d_loss'],
              'g_loss', out[tag + '_g_loss'], out[tag + '_g_loss_crop'], 'params', n_trainable())


# ------------------------------------------------------------------------------------------------ 6. conv_dropout wiring
def golden_dropout(out):
    """the reference's unet(..., conv_dropout=.4) in the learning phase (ext/neuron/models.py:320-324, 448-451): where the
    feature-wise Dropout layers sit, what the skip connections read (the conv layer's own output, models.py:431-432) and
    that the inference phase ignores them.  Every layer output and every drawn per-feature factor is stored."""
    rng = np.random.default_rng(8)
    x = rng.uniform(0, 1, (1, 16, 8, 16, 2)).astype(np.float32)
    for tag, phase in (('do_train', 1), ('do_infer', 0)):
        ks.reset(seed=6, learning_phase=phase)
        FEED[:] = [('unet_input', x)]
        model = nrn_models.unet(nb_features=8, input_shape=[16, 8, 16, 2], nb_levels=3, conv_size=3, nb_labels=1,
                                feat_mult=2, nb_conv_per_level=2, conv_dropout=.4, final_pred_activation='linear',
                                batch_norm=-1, activation='elu', input_model=None)
        out[tag + '_x'] = x
        out.update(params_dict(tag + '_w:'))
        out[tag + '_order'] = call_order()
        for layer, _, o in ks.GRAPH:  # the concatenations only: what the skip connections carry
            if 'merge' in str(layer.name) or 'concat' in str(layer.name):
                out['%s_act:%s' % (tag, layer.name)] = np.asarray(o)[0]
        out[tag + '_pred'] = np.asarray(model.output)[0]
        for k, v in ks.STATE['dropout'].items():
            assert v.shape[:4] == (1, 1, 1, 1), v.shape          # noise_shape [None, 1, 1, 1, C]: one factor per feature
            out['%s_scale:%s' % (tag, k)] = v.reshape(-1)
        for k, (m, v) in ks.STATE['bn_batch'].items():
            out['%s_bnmean:%s' % (tag, k)] = m
            out['%s_bnvar:%s' % (tag, k)] = v
        print(tag, 'pred', out[tag + '_pred'].shape, 'dropout layers', len(ks.STATE['dropout']),
              [n for n in call_order() if 'dropout' in n])


# ------------------------------------------------------------------------------------------------ 7. a batch of volumes
def golden_batch(out):
    """the reference's unet on a batch of TWO different volumes in the learning phase (training(batchsize=2)): the
    BatchNormalization layers' statistics run over batch and voxels, everything else is per volume"""
    rng = np.random.default_rng(9)
    x = rng.uniform(0, 1, (2, 16, 8, 16, 2)).astype(np.float32)
    x[1] *= 1.7
    ks.reset(seed=7, learning_phase=1)
    FEED[:] = [('unet_input', x)]
    model = nrn_models.unet(nb_features=8, input_shape=[16, 8, 16, 2], nb_levels=3, conv_size=3, nb_labels=1, feat_mult=2,
                            nb_conv_per_level=2, conv_dropout=0, final_pred_activation='linear', batch_norm=-1,
                            activation='elu', input_model=None)
    out['b2_x'] = x
    out.update(params_dict('b2_w:'))
    out['b2_pred'] = np.asarray(model.output)
    for k, (m, v) in ks.STATE['bn_batch'].items():
        out['b2_bnmean:%s' % k] = m
        out['b2_bnvar:%s' % k] = v
    print('batch pred', out['b2_pred'].shape)


def golden_batch_dropout(out):
    """batchsize 2 WITH conv_dropout: KL.Dropout(rate, noise_shape=[None, 1, 1, 1, C]) draws one keep mask per SAMPLE and
    feature (ext/neuron/models.py:320-324, 448-451); BatchNorm statistics over both samples of the dropped tensors"""
    rng = np.random.default_rng(19)
    x = rng.uniform(0, 1, (2, 16, 8, 16, 2)).astype(np.float32)
    x[1] *= 1.7
    ks.reset(seed=8, learning_phase=1)
    FEED[:] = [('unet_input', x)]
    model = nrn_models.unet(nb_features=8, input_shape=[16, 8, 16, 2], nb_levels=3, conv_size=3, nb_labels=1, feat_mult=2,
                            nb_conv_per_level=2, conv_dropout=.4, final_pred_activation='linear', batch_norm=-1,
                            activation='elu', input_model=None)
    out['b2d_x'] = x
    out.update(params_dict('b2d_w:'))
    out['b2d_pred'] = np.asarray(model.output)
    for k, v in ks.STATE['dropout'].items():
        assert v.shape[0] == 2 and v.shape[1:4] == (1, 1, 1), v.shape      # one factor per sample and feature
        out['b2d_scale:%s' % k] = v.reshape(2, -1)
    for k, (m, v) in ks.STATE['bn_batch'].items():
        out['b2d_bnmean:%s' % k] = m
        out['b2d_bnvar:%s' % k] = v
    differ = sum(int(not np.array_equal(v[0], v[1])) for k, v in out.items() if k.startswith('b2d_scale:'))
    print('batch + dropout pred', out['b2d_pred'].shape, 'dropout layers', len(ks.STATE['dropout']), 'with different masks per sample:', differ)


def golden_seg_loss_limits():
    """What the REFERENCE's own graph does for the two segmentation-loss configurations this build refuses with a ValueError:
    (a) two regression targets (`output_channel=[0, 1]`): training() builds the segmentation unet on `[..., 1]`
    (SynthSR/training.py:375) and add_seg_loss_to_model feeds it the whole `predicted_image` (metrics_model.py:149-163);
    (b) `target_res != atlas_res`: `segmentation_target` lives on the label maps' grid, the posteriors on the prediction's, and
    DiceLoss multiplies them voxel by voxel (metrics_model.py:187-207, ext/lab2im/layers.py:1300-1320).
    The glue is golden_seg_loss's; the outcome (exception type and message on the shim, or the loss value if the graph builds)
    goes to tests/golden/seg_loss_limits.json.  The shim checks shapes the way the ops it stands in for do (a convolution's
    kernel against its input's channel count, an element-wise product's operand shapes); it is not Keras, so the record shows
    THAT the reference graph breaks there, not the text Keras would print."""
    import json
    import keras.layers as KL
    rng = np.random.default_rng(7)
    res = {}
    gen_labels = np.array([0, 2, 3, 4, 41, 42, 43])
    seg_labels = np.array([0, 2, 3, 41, 42, 2, 77])
    for tag, S, S_seg, n_out in (('control_one_target_same_grid', [16, 16, 16], [16, 16, 16], 1),
                                 ('two_regression_targets', [16, 16, 16], [16, 16, 16], 2),
                                 ('segmentation_target_on_another_grid', [16, 16, 16], [8, 8, 8], 1)):
        image = rng.uniform(0, 1, [1] + S + [2]).astype(np.float32)
        target = rng.uniform(0, 1, [1] + S + [n_out]).astype(np.float32)
        seg = rng.integers(0, len(gen_labels), [1] + S_seg).astype(np.int32)
        ks.reset(seed=4)
        ks.STATE['frozen_bn_inference'] = False
        FEED[:] = [('gen_image', image), ('gen_target', target), ('gen_seg', seg[..., None])]
        im_in = KL.Input(shape=S + [2], name='gen_image')
        tg_in = KL.Input(shape=S + [n_out], name='gen_target')
        sg_in = KL.Input(shape=S_seg + [1], name='gen_seg', dtype='int32')
        seg_t = KL.Lambda(lambda x: x + 0, name='segmentation_target')(sg_in)
        tg = KL.Lambda(lambda x: x[0] + 0., name='regression_target')([tg_in, seg_t])
        im = KL.Lambda(lambda x: x[0] + 0., name='image_out')([im_in, tg])
        gen = models.Model(inputs=[im_in, tg_in, sg_in], outputs=[im, tg])
        rec = {'prediction_grid': S, 'segmentation_target_grid': S_seg, 'regression_targets': n_out}
        try:
            model = nrn_models.unet(nb_features=4, input_shape=S + [2], nb_levels=2, conv_size=3, nb_labels=n_out,
                                    feat_mult=2, nb_conv_per_level=2, conv_dropout=0, final_pred_activation='linear',
                                    batch_norm=-1, activation='elu', input_model=gen)
            model = models.Model(model.inputs, model.output)
            model = ref_mm.metrics_model(input_model=model, loss_cropping=None, metrics='l1', work_with_residual_channel=None)
            ks.PARAMS.clear()
            FEED[:] = [('unet_input', np.zeros([1] + S + [1], np.float32))]
            seg_model = nrn_models.unet(nb_features=4, input_shape=S + [1], nb_levels=2, conv_size=3,     # training.py:375
                                        nb_labels=len(seg_labels), feat_mult=2, nb_conv_per_level=2, conv_dropout=0,
                                        final_pred_activation='softmax', batch_norm=-1, activation='elu', input_model=None)
            total = ref_mm.add_seg_loss_to_model(input_model=model, seg_model=seg_model, generation_labels=gen_labels,
                                                 segmentation_label_equivalency=seg_labels, rel_weight=.25,
                                                 loss_cropping=None, m=None, M=None, fs_header=False)
            rec.update(raised=False, total_loss=float(np.asarray(total.outputs[0])))
        except Exception as e:   # noqa: BLE001 -- the outcome IS the record
            import traceback
            frames = traceback.extract_tb(e.__traceback__)
            where = [f for f in frames if '/root/reference/' in f.filename]
            rec.update(raised=True, type=type(e).__name__, message=str(e)[:300],
                       reference_frame=None if not where else '%s:%d' % (where[-1].filename.replace('/root/reference/', ''),
                                                                         where[-1].lineno))
        res[tag] = rec
        print(tag, rec)
    with open(os.path.join(OUT, 'seg_loss_limits.json'), 'w') as f:
        json.dump(res, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    if 'seg_limits' in sys.argv[1:]:
        golden_seg_loss_limits()
        sys.exit(0)
    which = sys.argv[1:] or ['wiring', 'graph', 'seg', 'critic', 'dropout', 'batch', 'batch_dropout']
    if 'batch_dropout' in which:
        out = {}
        golden_batch_dropout(out)
        np.savez_compressed(os.path.join(OUT, 'unet_batch_dropout.npz'), **out)
    if 'batch' in which:
        out = {}
        golden_batch(out)
        np.savez_compressed(os.path.join(OUT, 'unet_batch.npz'), **out)
    if 'dropout' in which:
        out = {}
        golden_dropout(out)
        np.savez_compressed(os.path.join(OUT, 'unet_dropout.npz'), **out)
    if 'wiring' in which:
        out = golden_param_count()
        golden_small_unet(out)
        np.savez_compressed(os.path.join(OUT, 'unet_wiring.npz'), **out)
    if 'graph' in which:
        out = {}
        golden_training_graph(out)
        np.savez_compressed(os.path.join(OUT, 'unet_training_graph.npz'), **out)
    if 'seg' in which:
        out = {}
        golden_seg_loss(out)
        np.savez_compressed(os.path.join(OUT, 'unet_seg_loss.npz'), **out)
    if 'critic' in which:
        out = {}
        golden_critic(out)
        np.savez_compressed(os.path.join(OUT, 'unet_critic.npz'), **out)
