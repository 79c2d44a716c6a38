"""Generates the golden vectors under tests/golden/ by executing the reference's own Python
source (read from /root/reference, never copied) on top of `tf_numpy_shim`.

Run in the development container only:   python tests/golden/gen/make_goldens.py
The resulting .npz files hold ONLY data: inputs, parameters, the random tape (raw N(0,1)/U[0,1)
draws in call order) and the reference's outputs.  See tests/golden/README.md for the list.
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

FEED = []
NAMED = shim.install(FEED, REF)

# reference modules (executed verbatim from /root/reference)
from ext.neuron import utils as nrn_utils  # noqa: E402
from ext.lab2im import utils as l2i_utils  # noqa: E402
from ext.lab2im import edit_tensors as l2i_et  # noqa: E402
from ext.lab2im import layers as l2i_layers  # noqa: E402
from ext.lab2im import edit_volumes as l2i_ev  # noqa: E402

_spec = importlib.util.spec_from_file_location('ref_l2i_model', os.path.join(REF, 'SynthSR', 'labels_to_image_model.py'))
ref_l2i_model = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ref_l2i_model)

from synthsr_amd.nifti import read_nifti  # noqa: E402

T = shim.T
t = shim.t


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


GEN_LABELS = np.load(os.path.join(REF, 'data', 'labels_classes_priors', 'generation_labels.npy'))
GEN_CLASSES = np.load(os.path.join(REF, 'data', 'labels_classes_priors', 'generation_classes.npy'))


def class_stats(rng, names=('t1_hr',)):
    means, stds = [], []
    for nm in names:
        pm = np.load(os.path.join(REF, 'data', 'labels_classes_priors', 'prior_means_%s.npy' % nm))
        ps = np.load(os.path.join(REF, 'data', 'labels_classes_priors', 'prior_stds_%s.npy' % nm))
        m = np.clip(rng.normal(pm[0], pm[1]), 0, None)[GEN_CLASSES]
        s = np.clip(rng.normal(ps[0], ps[1]), 0, None)[GEN_CLASSES]
        means.append(m)
        stds.append(s)
    return (np.stack(means, -1)[None].astype(np.float32), np.stack(stds, -1)[None].astype(np.float32))


# ----------------------------------------------------------------------------- unit goldens
def golden_resampler():
    """interpn / resize / integrate_vec / combine_non_linear_and_aff_to_shift (ext/neuron/utils.py)"""
    rng = np.random.default_rng(11)
    out = {}
    # resize linear 5^3x3 -> 16^3x3 (same code path as 5->80), and anisotropic 3x4x5 -> 12x10x9
    small = rng.standard_normal((5, 5, 5, 3)).astype(np.float32)
    out['rs_small'] = small
    out['rs_lin_16'] = np.asarray(nrn_utils.resize(t(small), [16 / 5] * 3, [16] * 3, 'linear'))
    small2 = rng.standard_normal((3, 4, 5, 2)).astype(np.float32)
    out['rs_small2'] = small2
    out['rs_lin_aniso'] = np.asarray(nrn_utils.resize(t(small2), [12 / 3, 10 / 4, 9 / 5], [12, 10, 9], 'linear'))
    # nearest down-resize 12x10x9 -> 8x4x3
    vol = rng.standard_normal((12, 10, 9, 1)).astype(np.float32)
    out['rs_vol'] = vol
    out['rs_near_down'] = np.asarray(nrn_utils.resize(t(vol), [8 / 12, 4 / 10, 3 / 9], [8, 4, 3], 'nearest'))
    # 1-D known answers quoted in SURVEY H6 / H16
    ramp = np.arange(5, dtype=np.float32)[:, None, None, None] * np.ones((1, 2, 2, 1), np.float32)
    out['rs_ramp80'] = np.asarray(nrn_utils.resize(t(ramp), [16., 1., 1.], [80, 2, 2], 'linear'))[:, 0, 0, 0]
    # integrate_vec (7 steps) on a 16^3 field
    vel = (3.0 * out['rs_lin_16']).astype(np.float32)
    out['iv_in'] = vel
    out['iv_out'] = np.asarray(nrn_utils.integrate_vec(t(vel), method='ss', nb_steps=7))
    # affine + elastic -> shift -> nearest / linear sampling
    aff = np.eye(4, dtype=np.float32)
    aff[:3, :3] += rng.uniform(-.1, .1, (3, 3)).astype(np.float32)
    aff[:3, 3] = rng.uniform(-2, 2, 3).astype(np.float32)
    field = out['iv_out']
    lab = load_label_crop(1, (60, 70, 60), (16, 16, 16)).astype(np.float32)[..., None]
    shift = nrn_utils.combine_non_linear_and_aff_to_shift([t(field), t(aff)], [16, 16, 16], shift_center=True)
    out['st_aff'] = aff
    out['st_field'] = field
    out['st_labels'] = lab
    out['st_shift'] = np.asarray(shift)
    out['st_nearest'] = np.asarray(nrn_utils.transform(t(lab), shift, 'nearest'))
    img = rng.standard_normal((16, 16, 16, 1)).astype(np.float32)
    out['st_img'] = img
    out['st_linear'] = np.asarray(nrn_utils.transform(t(img), shift, 'linear'))
    ashift = nrn_utils.affine_to_shift(t(aff), [16, 16, 16], shift_center=True)
    out['st_affine_only_shift'] = np.asarray(ashift)
    out['st_affine_only_linear'] = np.asarray(nrn_utils.transform(t(img), ashift, 'linear'))
    np.savez_compressed(os.path.join(OUT, 'resampler.npz'), **out)
    print('resampler.npz', {k: v.shape for k, v in out.items()})


def golden_host_math():
    """pure-numpy helpers + affine sampling + gaussian kernels"""
    out = {}
    # get_shapes cases: (labels_shape, output_shape, atlas_res, target_res, padding_margin, div_by_n)
    cases = [([160, 160, 160], None, [1., 1., 1.], [1., 1., 1.], None, 32),
             ([148, 187, 155], None, [1., 1., 1.], [1., 1., 1.], None, 32),
             ([148, 187, 155], 128, [1., 1., 1.], [1., 1., 1.], None, 32),
             ([148, 187, 155], [96, 128, 100], [1., 1., 1.], [1., 1., 1.], 8, 32),
             ([148, 187, 155], 160, [1., 1., 1.], [.7, .7, .7], None, 32),
             ([192, 192, 192], 192, [1., 1., 1.], [1., 1., 1.], None, None),
             ([40, 48, 36], 32, [1., 1., 1.], [1., 1., 1.], None, 32)]
    res = []
    for c in cases:
        crop, outs, pad = ref_l2i_model.get_shapes(list(c[0]), c[1], c[2], c[3], c[4], c[5])
        res.append(list(crop) + list(outs))
    out['get_shapes_in'] = np.array([[*c[0], -1 if c[1] is None else (c[1] if np.isscalar(c[1]) else -2),
                                      -1 if c[4] is None else c[4], -1 if c[5] is None else c[5]] for c in cases])
    out['get_shapes_out'] = np.array(res)
    # sample_affine_transform with a tape
    tape = shim.Tape(seed=5)
    shim.set_tape(tape)
    Tm = l2i_utils.sample_affine_transform(t(np.array([2], np.int32)), 3, rotation_bounds=15, scaling_bounds=.15,
                                           shearing_bounds=.02, translation_bounds=5)
    out['affine_T'] = np.asarray(Tm)
    out.update(tape_to_dict(tape, 'affine_tape'))
    # gaussian kernels
    for name, sig in [('k050', [.5] * 3), ('k042', [.42] * 3), ('khyp', [.63, .63, 2.1])]:
        out['gk_' + name] = np.asarray(l2i_et.gaussian_kernel(sig, separable=False))[..., 0, 0]
    tape = shim.Tape(seed=6)
    shim.set_tape(tape)
    out['gk_rand'] = np.asarray(l2i_et.gaussian_kernel([.42] * 3, blur_range=1.15, separable=False))[..., 0, 0]
    out.update(tape_to_dict(tape, 'gk_rand_tape'))
    # blurring sigma
    out['sigma_lr'] = l2i_et.blurring_sigma_for_downsampling([1., 1., 1.], [1.5, 1.5, 5.], .42, [1.5, 1.5, 5.])
    out['sigma_tgt'] = l2i_et.blurring_sigma_for_downsampling([1., 1., 1.], [1., 1., 1.])
    # mapping lut for a sided label list
    lab_list = np.array([0, 14, 15, 2, 3, 4, 41, 42, 43])
    out['swap_lut'] = l2i_utils.get_mapping_lut(lab_list, np.concatenate([lab_list[:3], lab_list[6:], lab_list[3:6]]))
    out['swap_lut_labels'] = lab_list
    np.savez_compressed(os.path.join(OUT, 'host_math.npz'), **out)
    print('host_math.npz', {k: getattr(v, 'shape', None) for k, v in out.items()})


def golden_layers():
    """call() bodies of the hot lab2im layers, each with its own tape (ext/lab2im/layers.py)"""
    out = {}
    rng = np.random.default_rng(21)
    lab = load_label_crop(2, (55, 80, 60), (24, 20, 28))[None, ..., None]
    means, stds = class_stats(rng, ('t1_hr', 't2'))
    # SampleConditionalGMM, 2 channels
    tape = shim.Tape(seed=31)
    shim.set_tape(tape)
    img = l2i_layers.SampleConditionalGMM(GEN_LABELS)([t(lab), t(means), t(stds)])
    out['gmm_labels'], out['gmm_means'], out['gmm_stds'], out['gmm_out'] = lab, means, stds, np.asarray(img)
    out.update(tape_to_dict(tape, 'gmm_tape'))
    # BiasFieldCorruption on one channel
    chan = np.asarray(img)[..., :1]
    tape = shim.Tape(seed=32)
    shim.set_tape(tape)
    b = l2i_layers.BiasFieldCorruption(.3, .25, False)(t(chan))
    out['bias_in'], out['bias_out'] = chan, np.asarray(b)
    out.update(tape_to_dict(tape, 'bias_tape'))
    # IntensityAugmentation
    tape = shim.Tape(seed=33)
    shim.set_tape(tape)
    ia = l2i_layers.IntensityAugmentation(clip=300, normalise=True, gamma_std=.5)(t(np.asarray(b)))
    out['ia_in'], out['ia_out'] = np.asarray(b), np.asarray(ia)
    out.update(tape_to_dict(tape, 'ia_tape'))
    # GaussianBlur fixed and randomised
    g1 = l2i_layers.GaussianBlur(sigma=.5)(t(np.asarray(ia)))
    out['blur_in'], out['blur_s050'] = np.asarray(ia), np.asarray(g1)
    tape = shim.Tape(seed=34)
    shim.set_tape(tape)
    g2 = l2i_layers.GaussianBlur([.63, .63, 2.1], 1.15)(t(np.asarray(g1)))
    out['blur_hyp_rand'] = np.asarray(g2)
    out.update(tape_to_dict(tape, 'blur_tape'))
    # RandomFlip with sided labels (swap active): list = [neutral(3) | left(3) | right(3)]
    lab_list = np.array([0, 14, 15, 2, 3, 4, 41, 42, 43])
    fl_in = lab_list[rng.integers(0, 9, (1, 6, 5, 4, 1))].astype(np.int32)
    for seed in (35, 39, 43):
        tape = shim.Tape(seed=seed)
        shim.set_tape(tape)
        fo = l2i_layers.RandomFlip(0, True, lab_list, 3)(t(fl_in))
        out['flip_out_%d' % seed] = np.asarray(fo)
        out.update(tape_to_dict(tape, 'flip_tape_%d' % seed))
    out['flip_in'], out['flip_label_list'] = fl_in, lab_list
    # resample_tensor with downsampling + reliability map (Hyperfine-like 1.5x1.5x5)
    vol = rng.standard_normal((1, 24, 24, 30, 1)).astype(np.float32)
    r, m = l2i_et.resample_tensor(t(vol), [24, 24, 30], 'linear', [1.5, 1.5, 5.], [1., 1., 1.], True)
    out['rt_in'], out['rt_out'], out['rt_map'] = vol, np.asarray(r), np.asarray(m)
    np.savez_compressed(os.path.join(OUT, 'layers.npz'), **out)
    print('layers.npz', {k: getattr(v, 'shape', None) for k, v in out.items()})


def run_graph(name, labels, means, stds, seed, real_image=None, aff=None, **kw):
    """whole labels_to_image_model() graph (SynthSR/labels_to_image_model.py:32-266)"""
    tape = shim.Tape(seed=seed)
    shim.set_tape(tape)
    NAMED.clear()
    FEED[:] = [('labels_input', labels), ('means_input', means), ('std_devs_input', stds)]
    if real_image is not None:
        FEED.append(('real_image_input', real_image))
    model = ref_l2i_model.labels_to_image_model(labels_shape=list(labels.shape[1:4]),
                                                generation_labels=GEN_LABELS,
                                                n_neutral_labels=len(GEN_LABELS),
                                                aff=np.eye(4) if aff is None else aff, **kw)
    image, target = model.outputs
    out = dict(labels=labels, means=means, stds=stds, image=np.asarray(image), target=np.asarray(target),
               seg=np.asarray(NAMED['segmentation_target']))
    if real_image is not None:
        out['real_image'] = real_image
    out.update(tape_to_dict(tape))
    for k, v in kw.items():
        if v is None:
            continue
        out['kw_' + k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, 'image', out['image'].shape, 'target', out['target'].shape, 'tape', len(tape.entries),
          [(k, a.shape) for k, a in tape.entries])


def golden_graphs():
    rng = np.random.default_rng(41)
    train_defaults = dict(atlas_res=[1., 1., 1.], target_res=None, output_div_by_n=32, padding_margin=None,
                          flipping=True, scaling_bounds=.15, rotation_bounds=15, shearing_bounds=.02,
                          translation_bounds=5, nonlin_std=4., nonlin_shape_factor=.03125 * 4,
                          simulate_registration_error=True, randomise_res=False, data_res=None, thickness=None,
                          downsample=True, build_reliability_maps=True, blur_range=1.15, bias_field_std=.3,
                          bias_shape_factor=.03125 * 4)
    # (a) config-2 structure at 32^3: 1 channel, training() defaults (shape factors x4 so the small grids are 4^3)
    lab = load_label_crop(1, (58, 78, 62), (32, 32, 32))[None, ..., None]
    means, stds = class_stats(rng)
    for seed in (101, 102, 103):
        run_graph('graph_c2_s%d' % seed, lab, means, stds, seed, input_channels=[True], output_channel=[0],
                  output_shape=32, **train_defaults)
    # (b) random crop: 40x48x36 labels cropped to 32^3
    lab_b = load_label_crop(3, (50, 70, 60), (40, 48, 36))[None, ..., None]
    run_graph('graph_crop_s111', lab_b, means, stds, 111, input_channels=[True], output_channel=[0],
              output_shape=32, **train_defaults)
    # (b-pad) PadAroundCentre (H3): padding_margin as training() sets it when loss_cropping is used (training.py:282-285),
    # scalar and per-axis; the padded 40^3 / 36x40x44 maps are then randomly cropped to 32^3; with a real-image target too
    run_graph('graph_pad_s161', lab, means, stds, 161, input_channels=[True], output_channel=[0], output_shape=32,
              **dict(train_defaults, padding_margin=4))
    run_graph('graph_pad_s162', lab, means, stds, 162, input_channels=[True], output_channel=[0], output_shape=32,
              **dict(train_defaults, padding_margin=[2, 4, 6]))
    # (b') real-image regression target (output_channel=None, tutorials 1/3/5): a smooth synthetic "scan" deformed,
    # cropped and flipped jointly with the labels (linear), then min-max normalised
    def fake_scan(lab_vol, seed):
        r = np.random.default_rng(seed)
        lut = r.uniform(20, 200, size=int(lab_vol.max()) + 1)
        v = lut[lab_vol[0, ..., 0]] + r.normal(0, 5, size=lab_vol.shape[1:4])
        return v.astype(np.float32)[None, ..., None]
    run_graph('graph_real_s131', lab, means, stds, 131, real_image=fake_scan(lab, 1), input_channels=[True],
              output_channel=None, output_shape=32, **train_defaults)
    run_graph('graph_real_crop_s132', lab_b, means, stds, 132, real_image=fake_scan(lab_b, 2), input_channels=[True],
              output_channel=None, output_shape=32, **train_defaults)
    run_graph('graph_real_pad_s163', lab, means, stds, 163, real_image=fake_scan(lab, 3), input_channels=[True],
              output_channel=None, output_shape=32, **dict(train_defaults, padding_margin=4))
    # (b'') randomise_res=True (fine_tuning_with_adversary defaults): SampleResolution -> separable 17-tap
    # DynamicGaussianBlur -> MimicAcquisition with distance map.  Keras' dynamic batch dimension is emulated for
    # edit_tensors.gaussian_kernel, which decides static-vs-batched sigma from `shape[0] is None` (edit_tensors.py:102-110)
    _orig_gk = l2i_et.gaussian_kernel

    def gk_none_batch(sigma, max_sigma=None, blur_range=None, separable=True):
        if isinstance(sigma, shim.T) and sigma.ndim == 2:
            sigma = sigma.view(shim.T)
            sigma._none_batch = True
        return _orig_gk(sigma, max_sigma, blur_range, separable)
    l2i_et.gaussian_kernel = gk_none_batch
    kw_rr = dict(train_defaults)
    kw_rr.update(randomise_res=True)
    for seed in (141, 142):
        run_graph('graph_rr_s%d' % seed, lab, means, stds, seed, input_channels=[True], output_channel=[0],
                  output_shape=32, **kw_rr)
    run_graph('graph_rr_crop_s143', lab_b, means, stds, 143, input_channels=[True], output_channel=[0],
              output_shape=32, **kw_rr)
    l2i_et.gaussian_kernel = _orig_gk
    # (c) Hyperfine-like: 3 channels [False, True, True], 1.5x1.5x5, registration error, no reliability maps
    means3, stds3 = class_stats(rng, ('t1_hr', 't1_lr', 't2'))
    kw = dict(train_defaults)
    kw.update(build_reliability_maps=False, data_res=np.array([[1.5, 1.5, 5.], [1.5, 1.5, 5.]]),
              thickness=np.array([[1.5, 1.5, 5.], [1.5, 1.5, 5.]]))
    run_graph('graph_hyperfine_s121', lab, means3, stds3, 121, input_channels=[False, True, True],
              output_channel=[0], output_shape=32, **kw)
    kw.update(build_reliability_maps=True)
    run_graph('graph_hyperfine_maps_s122', lab, means3, stds3, 122, input_channels=[False, True, True],
              output_channel=[0], output_shape=32, **kw)


def golden_inference():
    """host side of scripts/predict_command_line.py: edit_volumes.resample_volume (Gaussian pre-blur + linear
    RegularGridInterpolator on the reference's grid) and align_volume_to_ref, on small anisotropic volumes"""
    rng = np.random.RandomState(11)
    out = {}
    cases = {
        'down': (rng.rand(14, 11, 9) * 100, np.array([[0.7, 0, 0, -10.], [0, 0.8, 0, 5.], [0, 0, 0.6, 2.], [0, 0, 0, 1.]])),
        'up': (rng.rand(9, 8, 6) * 50, np.array([[1.5, 0, 0, 3.], [0, 1.5, 0, -4.], [0, 0, 5.0, 1.], [0, 0, 0, 1.]])),
        # oblique / permuted orientation: x <- -z, y <- x, z <- y with anisotropic voxels (down in one axis, up in two)
        'mixed': (rng.rand(10, 12, 7) * 80, np.array([[0, 0, -2.0, 30.], [0.9, 0, 0, -7.], [0, 1.3, 0, 4.], [0, 0, 0, 1.]])),
    }
    for name, (vol, aff) in cases.items():
        v2, a2 = l2i_ev.resample_volume(vol.copy(), aff.copy(), [1.0, 1.0, 1.0])
        v3, a3 = l2i_ev.align_volume_to_ref(v2, a2, aff_ref=np.eye(4), return_aff=True, n_dims=3)
        out[name + '_vol'], out[name + '_aff'] = vol, aff
        out[name + '_res_vol'], out[name + '_res_aff'] = v2, a2
        out[name + '_ras_vol'], out[name + '_ras_aff'] = v3, a3
    # resample_volume_like (predict_command_line_hyperfine.py:112): floating T2 resliced onto the (resampled, RAS) T1 grid
    flo = rng.rand(11, 9, 13) * 60
    # floating grid = the reference grid seen through a small rotation / anisotropic scaling / shift: partial overlap
    rel = np.array([[1.4, 0.1, 0, -2.5], [-0.1, 1.6, 0.05, 1.5], [0, 0.05, 1.2, -1.0], [0, 0, 0, 1.]])
    aff_flo = out['mixed_ras_aff'] @ rel
    out['like_flo'], out['like_flo_aff'] = flo, aff_flo
    out['like_out'] = l2i_ev.resample_volume_like(out['mixed_ras_vol'], out['mixed_ras_aff'], flo, aff_flo)
    np.savez_compressed(os.path.join(OUT, 'inference.npz'), **out)
    print('inference.npz', {k: getattr(v, 'shape', None) for k, v in out.items()})


def golden_separable_blur():
    """GaussianBlur separable branch (|sigma| > 5, ext/lab2im/layers.py:720,747-756 with edit_tensors.gaussian_kernel
    separable=True): fixed and blur_range-randomised, one axis with a 1-wide window (skipped), plus a whole
    labels_to_image_model graph of a 12 mm-thick-slice channel"""
    rng = np.random.default_rng(7)
    x = rng.uniform(0, 1, (1, 20, 18, 30, 1)).astype('float32')
    out = dict(blur_in=x)
    out['sep_fixed'] = np.asarray(l2i_layers.GaussianBlur(sigma=[1.0, 0.3, 5.2])(t(x)))     # 0.3 -> window 1: axis skipped
    tape = shim.Tape(seed=61)
    shim.set_tape(tape)
    out['sep_rand'] = np.asarray(l2i_layers.GaussianBlur([2.0, 1.1, 4.6], 1.15)(t(x)))
    out.update(tape_to_dict(tape, 'sep_tape'))
    np.savez_compressed(os.path.join(OUT, 'layers_separable.npz'), **out)
    lab = load_label_crop(1, origin=(64, 72, 58), shape=(32, 32, 32))[None, ..., None]
    means, stds = class_stats(np.random.default_rng(62))
    kw = dict(atlas_res=[1., 1., 1.], target_res=None, output_div_by_n=32, padding_margin=None, flipping=True,
              scaling_bounds=.15, rotation_bounds=15, shearing_bounds=.02, translation_bounds=5, nonlin_std=4.,
              nonlin_shape_factor=.125, simulate_registration_error=True, randomise_res=False, downsample=True,
              build_reliability_maps=True, blur_range=1.15, bias_field_std=.3, bias_shape_factor=.125)
    run_graph('graph_thick_s151', lab, means, stds, 151, input_channels=[True], output_channel=[0], output_shape=32,
              data_res=[1., 1., 12.5], thickness=[1., 1., 12.5], **kw)
    print('separable', out['sep_fixed'].shape)


def golden_nonras():
    """the whole graph with a NON-RAS `aff` and flipping (labels_to_image_model.py:154-162 hands RandomFlip the RAS axis of
    `aff`; the vendored RandomFlip reverses the axis at the POSITION of that entry in flip_axes, i.e. axis 0 whatever the
    affine says -- SURVEY F10): two tapes, one that flips and one that does not"""
    rng = np.random.default_rng(77)
    kw = dict(atlas_res=[1., 1., 1.], target_res=None, output_div_by_n=32, padding_margin=None, flipping=True,
              scaling_bounds=.15, rotation_bounds=15, shearing_bounds=.02, translation_bounds=5, nonlin_std=4.,
              nonlin_shape_factor=.125, simulate_registration_error=True, randomise_res=False, data_res=None, thickness=None,
              downsample=True, build_reliability_maps=True, blur_range=1.15, bias_field_std=.3, bias_shape_factor=.125)
    lab = load_label_crop(2, (60, 74, 60), (32, 32, 32))[None, ..., None]
    means, stds = class_stats(rng)
    aff = np.array([[0., 0., -1., 90.], [1., 0., 0., -126.], [0., -1., 0., 72.], [0., 0., 0., 1.]])   # PIL-like: x <- -k, y <- i, z <- -j
    for seed in (171, 175):
        run_graph('graph_nonras_s%d' % seed, lab, means, stds, seed, aff=aff, input_channels=[True], output_channel=[0],
                  output_shape=32, **kw)


def golden_metrics():
    """SynthSR/metrics_model.py:27-132 (`metrics_model`) run verbatim on a stand-in input model: l1 / l2 / laplace, with
    and without loss_cropping and work_with_residual_channel"""
    import types
    _s = importlib.util.spec_from_file_location('ref_metrics_model', os.path.join(REF, 'SynthSR', 'metrics_model.py'))
    ref_mm = importlib.util.module_from_spec(_s)
    _s.loader.exec_module(ref_mm)
    rng = np.random.RandomState(5)
    shape = (1, 14, 12, 11)  # last spatial size > 10: utils.get_dims reads a smaller one as a channel count
    out = dict(pred1=rng.standard_normal(shape + (1,)).astype('float32'),
               pred2=rng.standard_normal(shape + (2,)).astype('float32'),
               target=rng.uniform(0, 1, shape + (1,)).astype('float32'),
               image_out=rng.uniform(0, 1, shape + (2,)).astype('float32'),
               loss_cropping=np.array([8, 6, 4]))

    def run(metrics, crop, residual):
        layers_ = {'regression_target': t(out['target']), 'image_out': t(out['image_out'])}
        m = types.SimpleNamespace(outputs=[t(out['pred2' if metrics == 'laplace' else 'pred1'])], inputs=[],
                                  get_layer=lambda name: types.SimpleNamespace(output=layers_[name]))
        model = ref_mm.metrics_model(m, loss_cropping=crop, metrics=metrics, work_with_residual_channel=residual)
        return np.asarray(model.outputs, dtype='float64')

    for metrics in ('l1', 'l2', 'laplace'):
        out['loss_%s' % metrics] = run(metrics, None, None)
        out['loss_%s_crop' % metrics] = run(metrics, [8, 6, 4], None)
        out['loss_%s_res1' % metrics] = run(metrics, None, [1])
        out['loss_%s_crop_res1' % metrics] = run(metrics, [8, 6, 4], [1])
    np.savez_compressed(os.path.join(OUT, 'metrics.npz'), **out)
    print('metrics', {k: float(v) for k, v in out.items() if k.startswith('loss_') and v.ndim == 0})


def golden_estimate_priors():
    """SynthSR/estimate_priors.py:76-310 run verbatim on small synthetic datasets written as .npz volumes (the only
    format utils.load_volume reads without nibabel): two datasets, the second one with 2-channel images."""
    import tempfile
    _s = importlib.util.spec_from_file_location('ref_estimate_priors', os.path.join(REF, 'SynthSR', 'estimate_priors.py'))
    ref_ep = importlib.util.module_from_spec(_s)
    _s.loader.exec_module(ref_ep)
    rng = np.random.RandomState(77)
    labels_list = np.array([0, 2, 3, 4, 17, 41, 42, 53, 99])      # 99 never occurs
    classes_list = np.array([0, 1, 2, 3, 2, 1, 2, 2, 4])
    out = dict(labels_list=labels_list, classes_list=classes_list)
    tmp = tempfile.mkdtemp()
    dirs = []
    for d, (n_sub, n_ch) in enumerate([(3, 1), (2, 2)]):
        im_dir, la_dir = os.path.join(tmp, 'im%d' % d), os.path.join(tmp, 'la%d' % d)
        os.makedirs(im_dir), os.makedirs(la_dir)
        for i in range(n_sub):
            lab = load_label_crop(1 + (i % 2), origin=(50 + 7 * i, 60, 50 + 5 * d), shape=(24, 24, 24)).astype('int32')
            lab[lab == 41] = 41 if i else 2                       # a label missing from one subject
            chans = []
            for c in range(n_ch):
                mu = rng.uniform(20, 200, 256)
                im = mu[np.clip(lab, 0, 255)] + 12 * rng.standard_normal(lab.shape)
                im[rng.uniform(size=lab.shape) < 0.03] = 0        # zeros inside structures (keep_strictly_positive)
                im[lab == 0] *= (rng.uniform(size=lab.shape) < 0.5)[lab == 0]
                chans.append(im)
            im = (np.stack(chans, -1) if n_ch > 1 else chans[0]).astype('float32')
            np.savez_compressed(os.path.join(im_dir, 's%d.npz' % i), vol_data=im)
            np.savez_compressed(os.path.join(la_dir, 's%d.npz' % i), vol_data=lab)
            out['d%d_image_%d' % (d, i)] = im
            out['d%d_labels_%d' % (d, i)] = lab
        dirs.append((im_dir, la_dir))
    res_dir = os.path.join(tmp, 'res')
    pm, ps = ref_ep.build_intensity_stats([d[0] for d in dirs], [d[1] for d in dirs], res_dir, labels_list, classes_list)
    out['prior_means'], out['prior_stds'] = pm, ps
    assert np.array_equal(np.load(os.path.join(res_dir, 'prior_means.npy')), pm)
    pm1, ps1 = ref_ep.build_intensity_stats(dirs[0][0], dirs[0][1], res_dir, labels_list, None, rescale=False)
    out['prior_means_noclasses_norescale'], out['prior_stds_noclasses_norescale'] = pm1, ps1
    # single-image statistics, NaNs included (nanmedian / nan_policy='omit'), with and without the >0 filter
    im = out['d0_image_0'].copy()
    im[rng.uniform(size=im.shape) < 0.01] = np.nan
    out['single_image'] = im
    out['single_stats'] = ref_ep.sample_intensity_stats_from_image(im, out['d0_labels_0'], labels_list, classes_list)
    out['single_stats_keep0'] = ref_ep.sample_intensity_stats_from_image(im, out['d0_labels_0'], labels_list, classes_list,
                                                                         keep_strictly_positive=False)
    out['rescaled'] = l2i_ev.rescale_volume(out['d0_image_1'][:12])
    out['rescaled_pos'] = l2i_ev.rescale_volume(out['d0_image_1'][:12], new_min=-1, new_max=1, min_percentile=0,
                                                max_percentile=99, use_positive_only=True)
    np.savez_compressed(os.path.join(OUT, 'estimate_priors.npz'), **out)
    print('estimate_priors', pm.shape, ps.shape, pm1.shape)


def golden_gmm_batch():
    """SampleConditionalGMM on a BATCH of two (ext/lab2im/layers.py:480-498): the scatter indices are tiled over the batch and
    scattered into ONE look-up table, so the per-label means / stds of the two items are SUMMED and both items sample from
    the sums; two channels, different label crops and statistics per item"""
    out = {}
    rng = np.random.default_rng(23)
    lab = np.stack([load_label_crop(1, (60, 70, 60), (12, 10, 16)), load_label_crop(2, (55, 80, 60), (12, 10, 16))])[..., None]
    m0, s0 = class_stats(rng, ('t1_hr', 't2'))
    m1, s1 = class_stats(rng, ('t1_hr', 't2'))
    means, stds = np.concatenate([m0, m1], 0), np.concatenate([s0, s1], 0)
    tape = shim.Tape(seed=37)
    shim.set_tape(tape)
    img = l2i_layers.SampleConditionalGMM(GEN_LABELS)([t(lab), t(means), t(stds)])
    out['gmm2_labels'], out['gmm2_means'], out['gmm2_stds'], out['gmm2_out'] = lab, means, stds, np.asarray(img)
    out.update(tape_to_dict(tape, 'gmm2_tape'))
    np.savez_compressed(os.path.join(OUT, 'gmm_batch.npz'), **out)
    print('gmm_batch.npz', {k: getattr(v, 'shape', None) for k, v in out.items()})


if __name__ == '__main__':
    if 'gmm_batch' in sys.argv[1:]:
        golden_gmm_batch()
        sys.exit(0)
    which = sys.argv[1:] or ['resampler', 'host_math', 'layers', 'graphs', 'inference', 'estimate_priors', 'metrics', 'separable']
    if 'metrics' in which:
        golden_metrics()
    if 'separable' in which:
        golden_separable_blur()
    if 'estimate_priors' in which:
        golden_estimate_priors()
    if 'inference' in which:
        golden_inference()
    if 'resampler' in which:
        golden_resampler()
    if 'host_math' in which:
        golden_host_math()
    if 'layers' in which:
        golden_layers()
    if 'graphs' in which:
        golden_graphs()
    if 'nonras' in which or not sys.argv[1:]:
        golden_nonras()
