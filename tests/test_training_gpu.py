"""GPU end-to-end: the `training()` entry point on NIfTI label maps (checkpoint + resume), BrainGenerator.generate_brain,
and bench.py under torch.distributed.run with the RCCL gradient all-reduce path forced at world size 1."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _write_labels(tmp_path, n=2, shape=(40, 36, 48)):
    from synthsr_amd.nifti import write_nifti
    from synthsr_amd.synthetic import synthetic_label_map
    d = tmp_path / 'labels'
    d.mkdir()
    for i in range(n):
        write_nifti(str(d / ('brain%d_labels.nii.gz' % i)), synthetic_label_map(shape, 10 + i).astype(np.float32))
    return str(d)


def test_training_entry_point_checkpoint_and_resume(tmp_path):
    import torch
    from synthsr_amd.training import training
    from synthsr_amd.synthetic import GENERATION_LABELS, GENERATION_CLASSES, PRIOR_MEANS_T1_HR, PRIOR_STDS_T1_HR
    labels_dir = _write_labels(tmp_path)
    np.save(tmp_path / 'gl.npy', GENERATION_LABELS)
    np.save(tmp_path / 'gc.npy', GENERATION_CLASSES)
    np.save(tmp_path / 'pm.npy', PRIOR_MEANS_T1_HR)
    np.save(tmp_path / 'ps.npy', PRIOR_STDS_T1_HR)
    model_dir = str(tmp_path / 'models')
    kw = dict(path_generation_classes=str(tmp_path / 'gc.npy'), output_shape=32, n_levels=3, unet_feat_count=24,
              nonlin_shape_factor=.125, bias_shape_factor=.125, steps_per_epoch=3, verbose=False)
    net = training(labels_dir, model_dir, str(tmp_path / 'pm.npy'), str(tmp_path / 'ps.npy'), str(tmp_path / 'gl.npy'),
                   epochs=1, **kw)
    assert net.iterations == 3
    ck = os.path.join(model_dir, '001.npz')
    assert os.path.exists(ck)
    z = np.load(ck)
    assert 'unet_conv_downarm_0_0/kernel' in z.files and z['unet_conv_downarm_0_0/kernel'].shape == (3, 3, 3, 2, 24)
    assert 'unet_bn_down_0/moving_variance' in z.files and 'unet_likelihood/kernel' in z.files
    log = open(os.path.join(model_dir, 'logs', 'loss.csv')).read().strip().split('\n')
    assert len(log) == 1 and np.isfinite(float(log[0].split(',')[1]))
    # resume from the checkpoint: epoch parsed from the file name, optimizer state restored
    net2 = training(labels_dir, model_dir, str(tmp_path / 'pm.npy'), str(tmp_path / 'ps.npy'), str(tmp_path / 'gl.npy'),
                    epochs=2, checkpoint=ck, **kw)
    assert net2.iterations == 6 and os.path.exists(os.path.join(model_dir, '002.npz'))
    # the reference's checkpoint name, Keras layout: same weights as the .npz, and training resumes from it by name
    from synthsr_amd.keras_h5 import load_keras_weights
    h5 = load_keras_weights(os.path.join(model_dir, '001.h5'))
    assert all(np.array_equal(h5[k].reshape(z[k].shape), z[k]) for k in z.files if not k.startswith('optimizer/'))
    assert len(h5) == len([k for k in z.files if not k.startswith('optimizer/')])
    net3 = training(labels_dir, model_dir, str(tmp_path / 'pm.npy'), str(tmp_path / 'ps.npy'), str(tmp_path / 'gl.npy'),
                    epochs=2, checkpoint=os.path.join(model_dir, '001.h5'), **kw)
    assert net3.iterations == 3  # weights only: the optimizer restarts, like load_weights(by_name=True) in the reference
    # argument validation mirrors the reference's exceptions
    with pytest.raises(Exception):
        training(labels_dir, model_dir, None, None, str(tmp_path / 'gl.npy'), output_channel=None)
    with pytest.raises(Exception):
        training(labels_dir, model_dir, None, None, str(tmp_path / 'gl.npy'), output_channel=3)


def test_brain_generator_generate_brain(tmp_path):
    from synthsr_amd.brain_generator import BrainGenerator
    from synthsr_amd.synthetic import GENERATION_LABELS
    labels_dir = _write_labels(tmp_path, 1, (40, 36, 48))
    bg = BrainGenerator(labels_dir, None, None, 'uniform', GENERATION_LABELS, output_shape=32, output_div_by_n=8,
                        build_reliability_maps=True)
    image, target = bg.generate_brain()
    assert image.shape == (32, 32, 32, 2) and target.shape == (32, 32, 32)
    assert np.isfinite(image).all() and (image[..., 1] == 1).all() and 0 <= target.min() and target.max() <= 1 + 1e-6
    assert bg.model_output_shape == [32, 32, 32, 2] and bg.n_dims == 3 and bg.aff.shape == (4, 4)
    # padding margin (PadAroundCentre) and 2 input channels with registration error run through
    bg2 = BrainGenerator(labels_dir, None, None, 'uniform', GENERATION_LABELS, output_shape=32, output_div_by_n=8,
                         padding_margin=4, input_channels=[True, True], output_channel=1,
                         data_res=np.array([[1., 1., 3.], [1., 1., 1.]]), thickness=np.array([[1., 1., 3.], [1., 1., 1.]]),
                         downsample=True, build_reliability_maps=True)
    im2, tg2 = bg2.generate_brain()
    assert im2.shape == (32, 32, 32, 4) and np.isfinite(im2).all()
    assert set(np.unique(np.round(im2[..., 1], 6)).tolist()) != {1.0}  # channel 0 is down-sampled in z: sparse map


def test_real_image_targets_images_dir(tmp_path):
    """images_dir: BrainGenerator / training() with real scans as regression targets"""
    from synthsr_amd.brain_generator import BrainGenerator
    from synthsr_amd.nifti import write_nifti
    from synthsr_amd.synthetic import GENERATION_LABELS, synthetic_label_map
    from synthsr_amd.training import training
    labels_dir = _write_labels(tmp_path, 2, (40, 36, 48))
    d = tmp_path / 'images'
    d.mkdir()
    rng = np.random.RandomState(0)
    lut = rng.uniform(30, 220, 64)
    for i in range(2):  # a "scan" = intensity per label + noise, same grid as the label map
        lab = synthetic_label_map((40, 36, 48), 10 + i)
        write_nifti(str(d / ('brain%d.nii.gz' % i)), (lut[lab % 64] + rng.randn(*lab.shape)).astype(np.float32))
    bg = BrainGenerator(labels_dir, None, None, 'uniform', GENERATION_LABELS, images_dir=str(d), output_channel=None,
                        output_shape=32, output_div_by_n=8, build_reliability_maps=True)
    image, target = bg.generate_brain()
    assert image.shape == (32, 32, 32, 2) and target.shape == (32, 32, 32)
    assert np.isfinite(target).all() and target.min() == 0.0 and abs(target.max() - 1.0) < 1e-5
    assert len(np.unique(np.round(target, 3))) > 20  # a resampled scan, not a label map
    np.save(tmp_path / 'gl.npy', GENERATION_LABELS)
    net = training(labels_dir, str(tmp_path / 'models'), None, None, str(tmp_path / 'gl.npy'), images_dir=str(d),
                   output_channel=None, output_shape=32, n_levels=3, unet_feat_count=24, nonlin_shape_factor=.125,
                   bias_shape_factor=.125, steps_per_epoch=2, epochs=1, verbose=False)
    assert net.iterations == 2
    with pytest.raises(Exception, match='not both'):
        training(labels_dir, str(tmp_path / 'm2'), None, None, str(tmp_path / 'gl.npy'), images_dir=str(d), output_channel=0)


def test_training_with_segmentation_regularised_loss(tmp_path):
    """training(..., segmentation_model_file=...): frozen segmentation U-Net checkpoint, label list and
    equivalency from .npy files; loss = L1 + 0.25 * Dice"""
    import torch
    from synthsr_amd.training import training, save_checkpoint
    from synthsr_amd.synthetic import GENERATION_LABELS
    from synthsr_amd.unet import unet
    labels_dir = _write_labels(tmp_path, 2, (32, 32, 32))
    np.save(tmp_path / 'gl.npy', GENERATION_LABELS)
    seg_labels = np.array([0, 2, 3, 4, 41, 42, 17])
    np.save(tmp_path / 'seg_labels.npy', seg_labels)
    np.save(tmp_path / 'seg_eq.npy', np.array([0, 2, 3, 4, 2, 3, 17]))  # right-hemisphere labels merged onto the left ones
    seg_net = unet(24, [32, 32, 32, 1], 3, 3, len(seg_labels), feat_mult=2, nb_conv_per_level=2, batch_norm=-1,
                   activation='elu', final_pred_activation='softmax', seed=9)
    save_checkpoint(str(tmp_path / 'seg.npz'), seg_net)
    common = dict(output_shape=32, n_levels=3, unet_feat_count=24, nonlin_shape_factor=.125, bias_shape_factor=.125,
                  steps_per_epoch=2, epochs=1, verbose=False)
    net = training(labels_dir, str(tmp_path / 'm_seg'), None, None, str(tmp_path / 'gl.npy'),
                   segmentation_label_list=str(tmp_path / 'seg_labels.npy'),
                   segmentation_label_equivalency=str(tmp_path / 'seg_eq.npy'),
                   segmentation_model_file=str(tmp_path / 'seg.npz'), relative_weight_segmentation=0.25, **common)
    assert net.iterations == 2
    log = open(os.path.join(str(tmp_path / 'm_seg'), 'logs', 'loss.csv')).read().strip().split(',')
    total = float(log[1])
    net0 = training(labels_dir, str(tmp_path / 'm_plain'), None, None, str(tmp_path / 'gl.npy'), **common)
    plain = float(open(os.path.join(str(tmp_path / 'm_plain'), 'logs', 'loss.csv')).read().strip().split(',')[1])
    # same seeds, same first step: the regularised loss is the L1 loss plus 0.25 * Dice with 0 < Dice < 1
    assert np.isfinite(total) and plain < total < plain + 0.25


@pytest.mark.parametrize('metric,cropping', [('l2', 16), ('laplace', None), ('laplace', [24, 16, 16]), ('ssim', None),
                                             ('ssim', 24)])
def test_training_regression_metrics_and_loss_cropping(tmp_path, metric, cropping):
    """training(regression_metric='l2'|'laplace', loss_cropping=...) (SynthSR/training.py:85-87): the loss of a fixed
    batch goes down over the steps, the laplace network carries the 2-channel head, unknown metrics raise"""
    from synthsr_amd.training import training
    from synthsr_amd.synthetic import GENERATION_LABELS, GENERATION_CLASSES, PRIOR_MEANS_T1_HR, PRIOR_STDS_T1_HR
    labels_dir = _write_labels(tmp_path, n=1)
    for nm, a in (('gl', GENERATION_LABELS), ('gc', GENERATION_CLASSES), ('pm', PRIOR_MEANS_T1_HR), ('ps', PRIOR_STDS_T1_HR)):
        np.save(tmp_path / (nm + '.npy'), a)
    model_dir = str(tmp_path / 'models')
    args = (labels_dir, model_dir, str(tmp_path / 'pm.npy'), str(tmp_path / 'ps.npy'), str(tmp_path / 'gl.npy'))
    kw = dict(path_generation_classes=str(tmp_path / 'gc.npy'), output_shape=32, n_levels=3, unet_feat_count=24,
              nonlin_shape_factor=.125, bias_shape_factor=.125, steps_per_epoch=4, verbose=False, lr=1e-3)
    net = training(*args, epochs=3, regression_metric=metric, loss_cropping=cropping, **kw)
    assert net.nb_labels == (2 if metric == 'laplace' else 1) and net.iterations == 12
    z = np.load(os.path.join(model_dir, '003.npz'))
    assert z['unet_likelihood/kernel'].shape[-1] == net.nb_labels
    log = [float(l.split(',')[1]) for l in open(os.path.join(model_dir, 'logs', 'loss.csv')).read().strip().split('\n')]
    assert len(log) == 3 and all(np.isfinite(log)) and log[-1] < log[0]
    with pytest.raises(Exception):
        training(*args, regression_metric='huber', **kw)


def test_training_two_output_channels_with_residuals(tmp_path):
    """training(input_channels=[True, True], output_channel=[0, 1], work_with_residual_channel=[0, 1]) (SynthSR/
    training.py:246-249, metrics_model.py:53-64): 2-channel head, per-target residual channels"""
    from synthsr_amd.training import training
    from synthsr_amd.synthetic import GENERATION_LABELS, GENERATION_CLASSES, PRIOR_MEANS_T1_HR, PRIOR_STDS_T1_HR
    labels_dir = _write_labels(tmp_path, n=1)
    np.save(tmp_path / 'gl.npy', GENERATION_LABELS)
    np.save(tmp_path / 'gc.npy', GENERATION_CLASSES)
    np.save(tmp_path / 'pm.npy', np.concatenate([PRIOR_MEANS_T1_HR, PRIOR_MEANS_T1_HR[:, ::-1]]))
    np.save(tmp_path / 'ps.npy', np.concatenate([PRIOR_STDS_T1_HR, PRIOR_STDS_T1_HR]))
    model_dir = str(tmp_path / 'models')
    net = training(labels_dir, model_dir, str(tmp_path / 'pm.npy'), str(tmp_path / 'ps.npy'), str(tmp_path / 'gl.npy'),
                   path_generation_classes=str(tmp_path / 'gc.npy'), input_channels=[True, True], output_channel=[0, 1],
                   work_with_residual_channel=[0, 1], build_reliability_maps=False, output_shape=32, n_levels=3,
                   unet_feat_count=24, nonlin_shape_factor=.125, bias_shape_factor=.125, steps_per_epoch=4, epochs=3,
                   verbose=False, lr=1e-3)
    assert net.nb_labels == 2 and net.input_shape[3] == 2
    log = [float(l.split(',')[1]) for l in open(os.path.join(model_dir, 'logs', 'loss.csv')).read().strip().split('\n')]
    assert len(log) == 3 and all(np.isfinite(log)) and log[-1] < log[0]


def test_training_tutorial7_residual_channel_with_reliability_maps(tmp_path):
    """scripts/tutorials/7-training.py configuration: work_with_residual_channel=[0] together with build_reliability_maps
    (the reference repeats the list, SynthSR/training.py:270-271, F11): trains, and the residual is image_out[..., 0]"""
    from synthsr_amd.training import training
    from synthsr_amd.synthetic import GENERATION_LABELS, GENERATION_CLASSES, PRIOR_MEANS_T1_HR, PRIOR_STDS_T1_HR
    labels_dir = _write_labels(tmp_path, n=1)
    np.save(tmp_path / 'gl.npy', GENERATION_LABELS)
    np.save(tmp_path / 'gc.npy', GENERATION_CLASSES)
    np.save(tmp_path / 'pm.npy', PRIOR_MEANS_T1_HR)
    np.save(tmp_path / 'ps.npy', PRIOR_STDS_T1_HR)
    model_dir = str(tmp_path / 'models')
    kw = dict(path_generation_classes=str(tmp_path / 'gc.npy'), input_channels=[True], output_channel=[0],
              build_reliability_maps=True, data_res=[1., 1., 3.], output_shape=32, n_levels=3, unet_feat_count=24,
              nonlin_shape_factor=.125, bias_shape_factor=.125, steps_per_epoch=4, epochs=3, verbose=False, lr=1e-3)
    args = (labels_dir, model_dir, str(tmp_path / 'pm.npy'), str(tmp_path / 'ps.npy'), str(tmp_path / 'gl.npy'))
    net = training(*args, work_with_residual_channel=[0], **kw)
    assert net.input_shape[3] == 2
    log = [float(l.split(',')[1]) for l in open(os.path.join(model_dir, 'logs', 'loss.csv')).read().strip().split('\n')]
    assert len(log) == 3 and all(np.isfinite(log)) and log[-1] < log[0]
    kw2 = dict(kw, input_channels=[True, True], output_channel=[0, 1])
    with pytest.raises(ValueError):       # keras' Add cannot broadcast the repeated list [0, 1, 0, 1] against 2 channels
        training(*args, work_with_residual_channel=[0, 1], **kw2)


def test_bench_under_torchrun_with_forced_allreduce():
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(REPO, 'bench.py'), '--gpus', '1', '--steps', '3',
           '--warmup', '1', '--size', '64', '--no-cpu-baseline', '--force-allreduce']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['value'] > 0 and np.isfinite(d['final_loss'])
    assert d['roofline']['bound'] == 'mfma' and 0 < d['roofline']['frac'] < 1


@pytest.mark.parametrize('case', ['sr', 'synthesis', 'multimodal', 'real'])
def test_generation_examples_script(tmp_path, case):
    """scripts/tutorials/generate_examples.py (the use cases of the reference's tutorials 1-6) end to end on NIfTI files"""
    import importlib.util
    from synthsr_amd.nifti import write_nifti, read_nifti
    from synthsr_amd.synthetic import (GENERATION_LABELS, GENERATION_CLASSES, PRIOR_MEANS_T1_HR, PRIOR_STDS_T1_HR,
                                       synthetic_label_map)
    labels_dir = _write_labels(tmp_path, 2, (64, 64, 64))
    pri = tmp_path / 'priors'
    pri.mkdir()
    np.save(pri / 'generation_labels.npy', GENERATION_LABELS)
    np.save(pri / 'generation_classes.npy', GENERATION_CLASSES)
    for c in ('t1_hr', 't1_lr', 't2'):
        np.save(pri / ('prior_means_%s.npy' % c), PRIOR_MEANS_T1_HR)
        np.save(pri / ('prior_stds_%s.npy' % c), PRIOR_STDS_T1_HR)
    argv = [case, '--labels', labels_dir, '--priors', str(pri), '--out', str(tmp_path / 'out'), '-n', '2', '--shape', '32']
    if case == 'real':
        img = tmp_path / 'images'
        img.mkdir()
        for i in range(2):
            lab = synthetic_label_map((64, 64, 64), 10 + i)
            write_nifti(str(img / ('brain%d.nii.gz' % i)), (10.0 * (lab % 17) + 5).astype(np.float32))
        argv += ['--images', str(img)]
    spec = importlib.util.spec_from_file_location('gen_examples', os.path.join(REPO, 'scripts', 'tutorials',
                                                                              'generate_examples.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(argv)
    n_in = {'sr': 2, 'synthesis': 1, 'multimodal': 4, 'real': 2}[case]      # input channels (+ reliability maps)
    for i in range(2):
        im, _, _ = read_nifti(str(tmp_path / 'out' / ('image_%d.nii.gz' % i)))
        tg, _, _ = read_nifti(str(tmp_path / 'out' / ('target_%d.nii.gz' % i)))
        assert im.shape[:3] == (32, 32, 32) and (im.shape[3] if im.ndim == 4 else 1) == n_in and tg.shape == (32, 32, 32)
        assert np.isfinite(im).all() and np.isfinite(tg).all() and tg.max() > 0
