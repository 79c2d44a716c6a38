"""Host-side (numpy, float32) parameter math of the generator: everything that is O(1) per volume and
feeds the HIP kernels as small arguments — 4x4 affine sampling, blur-kernel weights, shape arithmetic,
look-up tables, reliability-map profiles.  The per-voxel work is in csrc/generator.hip.

float32 op order follows the reference graph (file:line cited per function) so that the kernels see the
same coefficients the reference's TF graph would compute; 4x4 products are k-ordered multiply/add
chains (TF's own accumulation order is third-party and unpinned).
"""
import math
import numpy as np

_F = np.float32


def _f(x):
    return np.asarray(x, dtype=np.float32)


def _mm(a, b):
    a, b = _f(a), _f(b)
    out = a[:, :1] * b[:1, :]
    for k in range(1, a.shape[1]):
        out = out + a[:, k:k + 1] * b[k:k + 1, :]
    return out


def reformat_to_list(var, length=None, dtype=None):
    """A parameter given as number / string / sequence / array as a python list (behaviour of ext/lab2im/utils.py:319-370
    for the value kinds this package passes): singletons are repeated to `length`, items cast to `dtype` ('int', 'float',
    'bool', 'str')."""
    if var is None:
        return None
    scalars = (int, float, bool, str, np.integer, np.floating, np.bool_)
    if isinstance(var, np.ndarray):
        var = var.item() if var.size == 1 else np.squeeze(var).tolist()
    if isinstance(var, scalars):
        items = [var]
    elif isinstance(var, (list, tuple)):
        items = list(var)
    else:
        raise TypeError('variable should be an int, float, bool, str, list, tuple or numpy array, had %s' % type(var))
    if length is not None and len(items) != length:
        if len(items) != 1:
            raise ValueError('if var is a list/tuple/numpy array, it should be of length 1 or %d, had %s' % (length, items))
        items = items * length
    if dtype is not None:
        cast = {'int': int, 'float': float, 'bool': bool, 'str': str}[dtype]
        items = [cast(v) for v in items]
    return items


def reformat_to_n_channels_array(var, n_dims=3, n_channels=1):
    """ext/lab2im/utils.py:373-407"""
    if var is None:
        return None
    if isinstance(var, (int, float, np.integer, np.floating)):
        return np.full((n_channels, n_dims), float(var))
    arr = np.array(var, dtype=np.float64)
    if arr.ndim <= 1:
        arr = arr.reshape(-1)
        if arr.size == 1:
            return np.full((n_channels, n_dims), float(arr[0]))
        if arr.size != n_dims:
            raise ValueError('if var is a list/tuple, it should be of length 1 or n_dims, had %s' % (var,))
        return np.tile(arr.reshape(1, n_dims), (n_channels, 1))
    if arr.shape != (n_channels, n_dims):
        raise ValueError('if array, var should be of shape (n_channels, n_dims), had %s' % (arr.shape,))
    return arr


def load_array_if_path(var, load_as_numpy=True):
    """ext/lab2im/utils.py:287-316"""
    if isinstance(var, str) and load_as_numpy:
        if not var.endswith(('.npy', '.npz')):
            raise ValueError('%s is not the path to a numpy array' % var)
        return np.load(var)
    return var


def find_closest_number_divisible_by_m(n, m):
    return n if n % m == 0 else int(n / m) * m


def get_resample_shape(shape, factor):
    """ext/lab2im/utils.py:577-588"""
    factor = reformat_to_list(factor, length=len(shape))
    return [math.ceil(shape[i] * factor[i]) for i in range(len(shape))]


def get_padding_margin(cropping, loss_cropping):
    """margin to pad the label maps with so that a centre box of `loss_cropping` inside a patch of `cropping` never sees
    padded borders: half the difference per axis (ext/lab2im/utils.py:601-614); an int if both are given as one number"""
    if cropping is None or loss_cropping is None:
        return None
    outer, inner = reformat_to_list(cropping), reformat_to_list(loss_cropping)
    n = max(len(outer), len(inner))
    outer, inner = reformat_to_list(outer, length=n), reformat_to_list(inner, length=n)
    margins = [int((o - i) / 2) for o, i in zip(outer, inner)]
    return margins if n > 1 else margins[0]


def get_shapes(labels_shape, output_shape, atlas_res, target_res, padding_margin, output_div_by_n):
    """(cropping_shape, output_shape, padding_margin) of the generator, the arithmetic of SynthSR/labels_to_image_model.py:
    269-335: the output is at most what the (padded) label map gives at the target resolution, floored to a multiple of
    `output_div_by_n`; the patch cropped from the label map is the output divided by the zoom (rounded)."""
    res_in, res_out = reformat_to_list(atlas_res), reformat_to_list(target_res)
    nd = len(res_in)
    size = [int(v) for v in labels_shape]
    if padding_margin is not None:
        padding_margin = reformat_to_list(padding_margin, length=nd, dtype='int')
        size = [v + 2 * m for v, m in zip(size, padding_margin)]
    zoom = None if res_in == res_out else [a / float(t) for a, t in zip(res_in, res_out)]

    def at_target_res(shape):
        return list(shape) if zoom is None else [int(v * z) for v, z in zip(shape, zoom)]

    def at_atlas_res(shape):
        return list(shape) if zoom is None else [int(np.around(v / z, 0)) for v, z in zip(shape, zoom)]

    def snapped(shape):
        return list(shape) if output_div_by_n is None else [find_closest_number_divisible_by_m(v, output_div_by_n)
                                                            for v in shape]
    largest = at_target_res(size)
    if output_shape is None:
        out = snapped(largest)
        crop = size if output_div_by_n is None else at_atlas_res(out)
        return crop, out, padding_margin
    wanted = reformat_to_list(output_shape, length=nd, dtype='int')
    out = [min(a, b) for a, b in zip(largest, wanted)]
    if snapped(out) != out:
        print('output shape {0} not divisible by {1}, changed to {2}'.format(out, output_div_by_n, snapped(out)))
        out = snapped(out)
    return at_atlas_res(out), out, padding_margin


def get_ras_axes(aff, n_dims=3):
    """for every image axis the world (R, A, S) axis it runs along: the largest |entry| of its column of inv(aff); if two
    image axes claim the same world axis, the later one takes the unclaimed axis (ext/lab2im/edit_volumes.py:591-606)"""
    weight = np.abs(np.linalg.inv(aff)[:n_dims, :n_dims])
    axes = weight.argmax(axis=0)
    for free in [i for i in range(n_dims) if i not in axes]:
        values, counts = np.unique(axes, return_counts=True)
        crowded = values[counts.argmax()]
        axes[np.flatnonzero(axes == crowded)[-1]] = free
    return axes


def uniform_f32(u, lo, hi):
    """tf.random.uniform(shape, minval, maxval) applied to a raw U[0,1) draw, float32"""
    lo, hi = _f(lo), _f(hi)
    return _f(u) * (hi - lo) + lo


def pick_bounds_block(b):
    """build-time half of utils.draw_value_from_distribution for an array hyperparameter (ext/lab2im/utils.py:1011-1016):
    a (2n, m) array is cut down to one of its n two-row blocks with np.random.randint (the reference's stream and call);
    everything else passes through (paths are loaded)."""
    b = load_array_if_path(b)
    if isinstance(b, np.ndarray):
        assert b.shape[0] % 2 == 0, 'number of rows of parameter_range should be divisible by 2'
        block = 2 * np.random.randint(b.shape[0] // 2)
        return b[block:block + 2, :]
    return b


def bounds_pair(b, centre, size):
    """hyperparameter -> (min[size], max[size]) — utils.draw_value_from_distribution (utils.py:1002-1016)"""
    b = load_array_if_path(b)
    if isinstance(b, np.ndarray):
        if b.shape[0] % 2 != 0:
            raise AssertionError('number of rows of parameter_range should be divisible by 2')
        if b.shape[0] != 2:  # direct callers; LabelsToImageModel resolves its arrays once with pick_bounds_block
            b = pick_bounds_block(b)
        return b[0].astype(np.float64), b[1].astype(np.float64)
    if b is None:
        raise ValueError('None bounds have caller-specific defaults; pass a number')
    if isinstance(b, (int, float, np.integer, np.floating)):
        return np.full(size, centre - b, np.float64), np.full(size, centre + b, np.float64)
    if isinstance(b, (list, tuple)):
        assert len(b) == 2, 'if list, parameter_range should be of length 2.'
        return np.full(size, b[0], np.float64), np.full(size, b[1], np.float64)
    raise ValueError('parameter_range should either be None, a number, a sequence, or a numpy array.')


def sample_affine(u, rotation_bounds=False, scaling_bounds=False, shearing_bounds=False, translation_bounds=False):
    """utils.sample_affine_transform (ext/lab2im/utils.py:675-752) for one item, 3-D.
    u: dict of raw uniforms {'rot':[3], 'shear':[6], 'scale':[3], 'trans':[3]} (only the enabled ones)."""
    eye = np.eye(3, dtype=_F)
    R = Sh = S = eye
    if rotation_bounds is not False:
        lo, hi = bounds_pair(rotation_bounds, 0., 3)
        r = uniform_f32(u['rot'], lo, hi) * _F(np.pi) / _F(180)  # utils.py:757
        c, s = np.cos(r), np.sin(r)
        Rx = _f([[1, 0, 0], [0, c[0], -s[0]], [0, s[0], c[0]]])
        Ry = _f([[c[1], 0, s[1]], [0, 1, 0], [-s[1], 0, c[1]]])
        Rz = _f([[c[2], -s[2], 0], [s[2], c[2], 0], [0, 0, 1]])
        R = _mm(_mm(Rx, Ry), Rz)  # :782
    if shearing_bounds is not False:
        lo, hi = bounds_pair(shearing_bounds, 0., 6)
        h = uniform_f32(u['shear'], lo, hi)
        Sh = _f([[1, h[0], h[1]], [h[2], 1, h[3]], [h[4], h[5], 1]])
    if scaling_bounds is not False:
        lo, hi = bounds_pair(scaling_bounds, 1., 3)
        S = np.diag(uniform_f32(u['scale'], lo, hi)).astype(_F)
    T = np.zeros((4, 4), _F)
    T[:3, :3] = _mm(S, _mm(Sh, R))  # :735
    if translation_bounds is not False:
        lo, hi = bounds_pair(translation_bounds, 0., 3)
        T[:3, 3] = uniform_f32(u['trans'], lo, hi)
    T[3, 3] = 1
    return T


def matmul4(a, b):
    return _mm(a, b)


def invert_affine(T):
    """tf.linalg.inv on a 4x4 (SynthSR/labels_to_image_model.py:207); float64 solve rounded to float32"""
    return np.linalg.inv(np.asarray(T, dtype=np.float64)).astype(_F)


def blurring_sigma_for_downsampling(current_res, downsample_res, mult_coef=None, thickness=None):
    """std dev (in voxels of `current_res`) of the Gaussian that mimics an acquisition at `downsample_res` with slices of
    `thickness`: coef * min(resolution, thickness) / current; coef .75 by default (.5 where nothing changes); 0 where the
    target resolution is 0 (ext/lab2im/edit_tensors.py:41-65, numpy branch)"""
    cur = np.asarray(current_res, dtype=np.float64)
    acq = np.array(downsample_res, dtype=np.float64)
    if thickness is not None:
        acq = np.minimum(acq, np.asarray(thickness, dtype=np.float64))
    if mult_coef is None:
        sigma = np.where(acq == cur, 0.5, 0.75 * acq / cur)
    else:
        sigma = mult_coef * acq / cur
    return np.where(acq == 0, 0.0, sigma)


def blur_window(sigma):
    """edit_tensors.py:124"""
    return [int(w) for w in (np.int32(np.ceil(2.5 * np.array(sigma, dtype=np.float64)) / 2) * 2 + 1)]


def gaussian_kernel(sigma, u_blur=None, blur_range=None):
    """et.gaussian_kernel, non-separable branch (ext/lab2im/edit_tensors.py:156-181), float32 [k0,k1,k2]"""
    sig = _f(sigma)
    if blur_range is not None and blur_range != 1:
        sig = sig * uniform_f32(u_blur, 1 / blur_range, blur_range)  # :121
    ws = blur_window(sigma)
    g = np.meshgrid(*[np.arange(w, dtype=_F) for w in ws], indexing='ij')
    diff = np.stack([g[d] - _F((ws[d] - 1) / 2) for d in range(3)], -1)
    s = sig.reshape(1, 1, 1, 3)
    zero = s == 0
    s1 = np.where(zero, _F(1), s)
    e = -np.square(diff) / (_F(2) * s1 ** 2)
    nrm = e - np.log(np.where(zero, _F(1), _F(np.sqrt(2 * np.pi)) * s))
    k = np.exp(np.sum(nrm, -1, dtype=_F))
    return (k / np.sum(k, dtype=_F)).astype(_F)


def gaussian_kernels_separable(sigma, u_blur=None, blur_range=None):
    """et.gaussian_kernel(..., separable=True) (ext/lab2im/edit_tensors.py:125-154) for a fixed sigma: per axis a float32
    1-D kernel exp(-d^2/2s^2 - log(sqrt(2 pi) s)) normalised to sum 1, or None where the window is 1 wide"""
    sig = _f(sigma)
    if blur_range is not None and blur_range != 1:
        sig = sig * uniform_f32(u_blur, 1 / blur_range, blur_range)
    out = []
    for a, w in enumerate(blur_window(sigma)):
        if w <= 1:
            out.append(None)
            continue
        loc = np.arange(w).astype(_F) - _F((w - 1) / 2)
        g = np.exp(-np.square(loc) / (_F(2) * sig[a] ** 2) - np.log(_F(np.sqrt(2 * np.pi)) * sig[a]))
        out.append((g / g.sum(dtype=_F)).astype(_F))
    return out


def is_separable_sigma(sigma):
    """GaussianBlur.build (ext/lab2im/layers.py:720)"""
    return bool(np.linalg.norm(np.array(sigma, dtype=np.float64)) > 5)


def reliability_profile(n_out, n_down):
    """one axis of the reliability map, float64 (ext/lab2im/edit_tensors.py:313-323)"""
    up = n_out / n_down
    loc = np.arange(0, n_out, up)
    fl = np.int32(np.floor(loc))
    ce = np.int32(np.clip(fl + 1, 0, n_out - 1))
    w = np.zeros(n_out)
    w[fl] = 1 - (loc - fl)
    w[ce] = w[ce] + (loc - fl)
    return w


def get_mapping_lut(source, dest=None):
    """int32 look-up table lut[source[i]] = dest[i] (default dest = 0..n-1), zeros elsewhere (ext/lab2im/utils.py:894-914)"""
    keys = np.asarray(reformat_to_list(source), dtype=np.int32)
    if dest is None:
        values = np.arange(len(keys), dtype=np.int32)
    else:
        assert len(source) == len(dest), 'label_list and new_label_list should have the same length'
        values = np.asarray(reformat_to_list(dest, dtype='int'))
    lut = np.zeros(int(keys.max()) + 1, dtype=np.int32)
    lut[keys] = values            # later duplicates win, as in an element-by-element fill
    return lut


def flip_swap_lut(label_list, n_neutral_labels):
    """RandomFlip.build (ext/lab2im/layers.py:375-386); None when there are no sided labels"""
    label_list = np.asarray(label_list)
    n = len(label_list)
    if n_neutral_labels is None or n_neutral_labels == n:
        return None
    split = np.split(label_list, [n_neutral_labels, n_neutral_labels + int((n - n_neutral_labels) / 2)])
    return get_mapping_lut(label_list, np.concatenate((split[0], split[2], split[1])))


def gmm_luts(generation_labels, means, stds):
    """SampleConditionalGMM's scatter_nd LUTs for one item (ext/lab2im/layers.py:472-495) -> [2, C, L] float32.
    tf.scatter_nd accumulates duplicate labels; so does np.add.at."""
    generation_labels = np.asarray(generation_labels).astype(np.int64)
    means, stds = _f(means), _f(stds)
    C = means.shape[-1]
    L = int(generation_labels.max()) + 1
    lut = np.zeros((2, C, L), _F)
    for c in range(C):
        np.add.at(lut[0, c], generation_labels, means[:, c])
        np.add.at(lut[1, c], generation_labels, stds[:, c])
    return lut


def batch_gmm_parameters(means, stds, sum_over_batch=False):
    """per-item GMM parameters of a batch: means / stds [B, L, C] -> two lists of B arrays [L, C].
    sum_over_batch: what the reference's SampleConditionalGMM does to a batch -- its scatter indices are tiled over the batch
    and scattered into ONE look-up table, so every item samples from the SUM of the B items' means / stds
    (ext/lab2im/layers.py:482-495; SURVEY F9, tests/golden/gmm_batch.npz).  Identical for B = 1."""
    means, stds = _f(means), _f(stds)
    B = means.shape[0]
    if sum_over_batch and B > 1:
        m, s = means.sum(0, dtype=_F), stds.sum(0, dtype=_F)
        return [m] * B, [s] * B
    return [means[b] for b in range(B)], [stds[b] for b in range(B)]


def draw_value_from_distribution(hyperparameter, size=1, distribution='uniform', centre=0., default_range=10.0,
                                 positive_only=False, rng=None):
    """`size` values from U(a, b) or N(a, b) with (a, b) given as in ext/lab2im/utils.py:961-1049 (numpy branch): None ->
    centre -/+ default_range; a number r -> centre -/+ r; a pair -> (a, b) for every value; an array [2 m, size] -> one of
    its m two-row blocks, picked at random, row 0 = a, row 1 = b per value.  False -> None (feature switched off)."""
    if hyperparameter is False:
        return None
    source = np.random if rng is None else rng
    hp = load_array_if_path(hyperparameter)
    if isinstance(hp, np.ndarray):
        assert hp.shape[0] % 2 == 0, 'number of rows of parameter_range should be divisible by 2'
        n_blocks = hp.shape[0] // 2
        block = int(source.randint(n_blocks) if hasattr(source, 'randint') else source.integers(n_blocks))
        first, second = hp[2 * block, :], hp[2 * block + 1, :]
    elif hp is None or isinstance(hp, (int, float)):
        half = default_range if hp is None else hp
        first, second = np.full(size, centre - half), np.full(size, centre + half)
    elif isinstance(hp, (list, tuple)):
        assert len(hp) == 2, 'if list, parameter_range should be of length 2.'
        first, second = np.repeat(hp[0], size), np.repeat(hp[1], size)
    else:
        raise ValueError('parameter_range should either be None, a number, a sequence, or a numpy array.')
    if distribution == 'uniform':
        value = source.uniform(low=first, high=second)
    elif distribution == 'normal':
        value = source.normal(loc=first, scale=second)
    else:
        raise ValueError("Distribution not supported, should be 'uniform' or 'normal'.")
    return np.maximum(value, 0) if positive_only else value


def randomise_res_plan(u_rr, u_blur, blur_range, atlas_res, crop_shape, output_shape, max_res=9.0, prob_min=0.05):
    """Host part of the randomise_res path (SynthSR/labels_to_image_model.py:215-220) for one channel, float32 like the
    reference graph: SampleResolution (ext/lab2im/layers.py:598-652, isotropic-only branch: resolution ~ U(atlas, 9) per
    axis, 5 % chance of the atlas resolution, thickness ~ U(atlas, resolution)), the blur sigma
    (edit_tensors.py:67-82: .42 * min(resolution, thickness) / atlas), the three 1-D kernels of DynamicGaussianBlur
    (edit_tensors.py:113-154; width from max_sigma = .75 * 9 / atlas) and MimicAcquisition's zoom factors
    (layers.py:943-946).  u_rr = (axis pick [unused by this branch], u_res[3], u_gate, u_thick[3]) raw uniforms."""
    f = np.float32
    _, u_res, u_gate, u_thick = u_rr
    lo = np.asarray(atlas_res, dtype=f)
    hi = np.full(3, max_res, dtype=f)
    res = np.asarray(u_res, dtype=f) * (hi - lo) + lo
    if f(np.asarray(u_gate, dtype=f).reshape(-1)[0]) < f(prob_min):
        res = lo.copy()
    thick = np.asarray(u_thick, dtype=f) * (res - lo) + lo
    dres = np.minimum(res, thick)
    sig = np.where(dres == 0, f(0), f(.42) * dres / lo).astype(f)
    if blur_range is not None and blur_range != 1:
        sig = sig * (np.asarray(u_blur, dtype=f) * (f(blur_range) - f(1 / blur_range)) + f(1 / blur_range))
    max_sigma = 0.75 * max_res / np.asarray(atlas_res, dtype=np.float64)
    win = (np.int32(np.ceil(2.5 * max_sigma) / 2) * 2 + 1).tolist()
    kernels = []
    for a in range(3):
        loc = np.arange(win[a]).astype(f) - f((win[a] - 1) / 2)
        e = -np.square(loc) / (f(2) * sig[a] ** 2)
        g = np.exp(e - np.log(f(np.sqrt(2 * np.pi)) * sig[a]))
        kernels.append((g / g.sum(dtype=f)).astype(f))
    S = np.asarray(crop_shape)
    down_shape = (np.asarray(S * np.asarray(atlas_res, dtype=np.float64), dtype=f) / res).astype(np.int32)
    down_zoom = (down_shape / S).astype(f)
    up_zoom = (np.asarray(output_shape, dtype=np.int32) / down_shape).astype(f)
    return dict(res=[float(v) for v in res], thickness=[float(v) for v in thick], sigma=[float(v) for v in sig],
                kernels=kernels, down_zoom=[float(v) for v in down_zoom], up_zoom=[float(v) for v in up_zoom])
