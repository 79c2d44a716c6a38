"""Inference path of the reference's `scripts/predict_command_line.py:58-138`: resample to 1 mm,
re-orient to RAS, min-max normalise, zero-pad to a multiple of 32, U-Net forward on the MI355X (optionally averaged
with the left-right flipped pass), rescale / clip, crop, save.  Host steps are numpy like the reference's; the U-Net
runs through the HIP kernels (`UNet3D.predict`, inference-mode BatchNorm with the moving statistics).

The reference ships its weights as a Keras `.h5` (`models/SynthSR_v10_210712.h5`, not part of the reference
checkout); `--model` / `path_model` take such a file (read by the library-free HDF5 reader of keras_h5.py) or the `.npz`
written by `synthsr_amd.training.save_checkpoint`, whose keys are the same Keras layer names.
"""
import os
import numpy as np

from . import volumes as V

DEFAULT_MODEL_HYPERFINE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'models',
                                       'SynthSR_v10_210712_hyperfine.npz')
DEFAULT_MODEL = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'models', 'SynthSR_v10_210712.npz')


def prepare_volume(im, aff, ct=False):
    """predict_command_line.py:110-124.  Returns (padded [W0,W1,W2] float64, crop offsets, unpadded shape, RAS affine)"""
    im = np.array(im, dtype=np.float64)
    if ct:
        im[im < 0] = 0
        im[im > 80] = 80
    im, aff = V.resample_volume(im, aff, [1.0, 1.0, 1.0])
    im, aff2 = V.align_volume_to_ref(im, aff, aff_ref=np.eye(4), return_aff=True, n_dims=3)
    im = im - np.min(im)
    im = im / np.max(im)
    shape = np.array(im.shape)
    W = (np.ceil(shape / 32.0) * 32).astype('int')
    idx = np.floor((W - shape) / 2).astype('int')
    S = np.zeros(W)
    S[idx[0]:idx[0] + shape[0], idx[1]:idx[1] + shape[1], idx[2]:idx[2] + shape[2]] = im
    return S, idx, shape, aff2


def postprocess(output, idx, shape):
    """predict_command_line.py:131-135: intensities to [0, 128] of a 255 scale, crop the padding"""
    pred = 255 * np.squeeze(np.asarray(output, dtype=np.float64))
    pred[pred < 0] = 0
    pred[pred > 128] = 128
    return pred[idx[0]:idx[0] + shape[0], idx[1]:idx[1] + shape[1], idx[2]:idx[2] + shape[2]]


class Predictor:
    """The U-Net of predict_command_line.py:65-76 (24 features, 5 levels, 2 convs per level, 1 input channel), rebuilt
    per padded volume shape (the reference's Keras graph has `None` spatial dims; weights are shape-independent)."""

    def __init__(self, path_model=None, device=None, state_dict=None, n_inputs=1):
        import torch
        self.torch = torch
        self.device = device or 'cuda'
        self.state = state_dict
        self.n_inputs = n_inputs  # 1: predict_command_line.py; 2 (T1, T2): predict_command_line_hyperfine.py:60-71
        if state_dict is None:
            if path_model is None:
                # the reference's own file name first (scripts/predict_command_line.py:79), then our .npz layout
                default = DEFAULT_MODEL if n_inputs == 1 else DEFAULT_MODEL_HYPERFINE
                path_model = next((p for p in (default[:-4] + '.h5', default) if os.path.isfile(p)), default)
            if not os.path.isfile(path_model):
                raise FileNotFoundError('model file %s not found (a Keras .h5 as distributed by the reference, or the '
                                        '.npz written by synthsr_amd.training.save_checkpoint)' % path_model)
            from .training import read_weights
            z = read_weights(path_model)
            self.state = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in z.items()
                          if not k.startswith('optimizer/')}
        self.nets = {}

    def net_for(self, shape):
        from .unet import unet
        key = tuple(int(s) for s in shape)
        if key not in self.nets:
            self.nets.clear()  # one resident network: volumes of a folder usually share their shape
            net = unet(nb_features=24, input_shape=list(key) + [self.n_inputs], nb_levels=5, conv_size=3, nb_labels=1, feat_mult=2,
                       nb_conv_per_level=2, conv_dropout=0, final_pred_activation='linear', batch_norm=-1,
                       activation='elu', input_model=None, device=self.device)
            # a weight file with another layer prefix / missing layers must not leave random weights in place silently
            missing = [nm for nm, _, _ in net.specs if nm not in self.state]
            missing += [b['name'] + '/moving_mean' for b in net.bn_layers if b['name'] + '/moving_mean' not in self.state]
            if missing:
                raise KeyError('the weight file lacks %d of the network\'s tensors (first: %s): wrong model file or layer '
                               'prefix?' % (len(missing), missing[0]))
            net.load_state_dict(self.state, strict=True)
            net.repack()
            self.nets[key] = net
        return self.nets[key]

    def __call__(self, S, flipping=True):
        """S [W0,W1,W2] or [W0,W1,W2,n_inputs] (numpy) -> U-Net output [W0,W1,W2] float32 (numpy); flipping: average
        with the pass on the volume flipped along the first (left-right) axis"""
        torch = self.torch
        net = self.net_for(S.shape[:3])
        x = torch.from_numpy(np.ascontiguousarray(S, dtype=np.float32)).to(self.device)
        if x.dim() == 3:
            x = x[..., None]
        x = x.contiguous()
        out = net.predict(x).clone()[..., 0]
        if flipping:
            outf = net.predict(torch.flip(x, dims=[0]).contiguous())[..., 0]
            out = 0.5 * out + 0.5 * torch.flip(outf, dims=[0])
        return out.cpu().numpy()


def predict(path_images, path_predictions, path_model=None, ct=False, disable_flipping=False, device=None, verbose=True,
            predictor=None):
    """Same contract as the reference script: `path_images` / `path_predictions` are both single files (.nii, .nii.gz,
    .npz / .mgz) or both folders."""
    path_images = os.path.abspath(path_images)
    basename = os.path.basename(path_images)
    path_predictions = os.path.abspath(path_predictions)
    if not any(ext in basename for ext in ('.nii.gz', '.nii', '.mgz', '.npz')):
        if os.path.isfile(path_images):
            raise Exception('extension not supported for %s, only use: nii.gz, .nii, .mgz, or .npz' % path_images)
        images = V.list_images_in_folder(path_images)
        os.makedirs(path_predictions, exist_ok=True)
        outs = [os.path.join(path_predictions, os.path.basename(p)).replace('.nii', '_SynthSR.nii') for p in images]
        outs = [p.replace('.mgz', '_SynthSR.mgz').replace('.npz', '_SynthSR.npz') for p in outs]
    else:
        assert os.path.isfile(path_images), "files does not exist: %s " \
                                            "\nplease make sure the path and the extension are correct" % path_images
        images, outs = [path_images], [path_predictions]
    predictor = predictor or Predictor(path_model, device)
    if verbose:
        print('Found %d images' % len(images))
    for i, (pin, pout) in enumerate(zip(images, outs)):
        if verbose:
            print('  Working on image %d ' % (i + 1))
            print('  ' + pin)
        im, aff, _ = V.load_volume(pin, im_only=False, dtype='float')
        S, idx, shape, aff2 = prepare_volume(im, aff, ct=ct)
        out = predictor(S, flipping=not disable_flipping)
        V.save_volume(postprocess(out, idx, shape), aff2, None, pout)
    return outs


def prepare_hyperfine(im1, aff1, im2, aff2):
    """predict_command_line_hyperfine.py:105-126: T1 to 1 mm RAS, T2 resliced onto it, the script's intensity scalings,
    zero-pad to a multiple of 32.  Returns (S [W,2], idx, shape, affine, (minimum, spread), normalised T1)"""
    im1, aff1 = V.resample_volume(np.array(im1, dtype=np.float64), aff1, [1.0, 1.0, 1.0])
    im1, aff1_mod = V.align_volume_to_ref(im1, aff1, aff_ref=np.eye(4), return_aff=True, n_dims=3)
    im2 = V.resample_volume_like(im1, aff1_mod, np.array(im2, dtype=np.float64), aff2)
    minimum = np.min(im1)
    im1 = im1 - minimum
    spread = np.max(im1) / 3.0  # the reference's training-time scaling (:116)
    im1 = im1 / spread
    im2 = im2 - np.min(im2)
    im2 = im2 / np.max(im2) * 2.0  # (:119)
    I = np.stack([im1, im2], axis=-1)
    shape = np.array(I.shape[:3])
    W = (np.ceil(shape / 32.0) * 32).astype('int')
    idx = np.floor((W - shape) / 2).astype('int')
    S = np.zeros([*W, 2])
    S[idx[0]:idx[0] + shape[0], idx[1]:idx[1] + shape[1], idx[2]:idx[2] + shape[2], :] = I
    return S, idx, shape, aff1_mod, (minimum, spread), im1


def postprocess_hyperfine(output, idx, shape, scaling, im1):
    """:129-131: the network predicts a residual on the normalised T1"""
    minimum, spread = scaling
    res = np.squeeze(np.asarray(output, dtype=np.float64))[idx[0]:idx[0] + shape[0], idx[1]:idx[1] + shape[1],
                                                             idx[2]:idx[2] + shape[2]]
    pred = minimum + spread * (res + im1)
    pred[pred < 0] = 0
    return pred


def predict_hyperfine(path_t1_images, path_t2_images, path_predictions, path_model=None, device=None, verbose=True,
                      predictor=None):
    """scripts/predict_command_line_hyperfine.py: pairs of (T1, T2) low-field scans -> synthetic 1 mm MP-RAGE.  All three
    paths are files, or all three are folders (T1 / T2 folders sorted alike)."""
    path_t1_images, path_t2_images = os.path.abspath(path_t1_images), os.path.abspath(path_t2_images)
    path_predictions = os.path.abspath(path_predictions)
    b1 = os.path.basename(path_t1_images)
    if not any(ext in b1 for ext in ('.nii.gz', '.nii', '.mgz', '.npz')):
        if os.path.isfile(path_t1_images):
            raise Exception('extension not supported for %s, only use: nii.gz, .nii, .mgz, or .npz' % path_t1_images)
        t1s, t2s = V.list_images_in_folder(path_t1_images), V.list_images_in_folder(path_t2_images)
        os.makedirs(path_predictions, exist_ok=True)
        outs = [os.path.join(path_predictions, os.path.basename(p)).replace('.nii', '_SynthSR.nii') for p in t1s]
        outs = [p.replace('.mgz', '_SynthSR.mgz').replace('.npz', '_SynthSR.npz') for p in outs]
    else:
        assert os.path.isfile(path_t1_images), "files does not exist: %s " \
                                               "\nplease make sure the path and the extension are correct" % path_t1_images
        t1s, t2s, outs = [path_t1_images], [path_t2_images], [path_predictions]
    predictor = predictor or Predictor(path_model, device, n_inputs=2)
    if verbose:
        print('Found %d images' % len(t1s))
    for i, (p1, p2, pout) in enumerate(zip(t1s, t2s, outs)):
        if verbose:
            print('  Working on image %d ' % (i + 1))
            print('  ' + p1 + ', ' + p2)
        im1, aff1, _ = V.load_volume(p1, im_only=False, dtype='float')
        im2, aff2, _ = V.load_volume(p2, im_only=False, dtype='float')
        S, idx, shape, aff_out, scaling, t1n = prepare_hyperfine(im1, aff1, im2, aff2)
        out = predictor(S, flipping=False)  # the Hyperfine script has no flip averaging
        V.save_volume(postprocess_hyperfine(out, idx, shape, scaling, t1n), aff_out, None, pout)
    return outs
