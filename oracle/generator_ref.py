"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement in numpy/float32 of the reference's synthetic-brain generator
(SynthSR/labels_to_image_model.py:32-266 and the lab2im / neuron functions it calls).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

PARITY PINNING: every function below is checked in tests/test_oracle_golden.py against golden
vectors produced by executing the reference's own source on a numpy TF shim
(tests/golden/gen/make_goldens.py, development container only).  Integer outputs (deformed label
maps, LUT swaps, nearest resizes) are bit-exact; float32 outputs agree to the tolerances written in
that test.  Third-party TF/Keras arithmetic that is not in /root/reference (matmul accumulation
order, conv3d accumulation order, exp/pow/cos rounding, RNG streams) is UNPINNED: the conventions
used here are k-ordered float32 multiply/add without FMA for the 4x4 matmuls, (z,y,x)-raster float32
accumulation for the blur, numpy float32 libm for transcendentals, and explicit injected draws.

All random inputs are explicit: raw U[0,1) / N(0,1) draws ("tape" entries) are arguments.
Everything is float32 unless noted; 3-D volumes only (the product is 3-D only).
"""
import itertools
import math
import numpy as np

F = np.float32


def f32(x):
    return np.asarray(x, dtype=np.float32)


# =============================================================================================
# resampler core — ext/neuron/utils.py
# =============================================================================================
def _grid(shape):
    """float32 ndgrid, 'ij' indexing (ext/neuron/utils.py:389-446)"""
    return [g.astype(F) for g in np.meshgrid(*[np.arange(s) for s in shape], indexing='ij')]


def interpn(vol, loc, method='linear'):
    """ext/neuron/utils.py:25-124.  vol [X,Y,Z,C] (any dtype), loc [..., 3] float32"""
    vol = np.asarray(vol)
    loc = f32(loc)
    S = vol.shape[:3]
    flat = vol.reshape(-1, vol.shape[-1])
    if method == 'nearest':
        r = np.round(loc).astype(np.int32)  # half-to-even, :114
        idx = [np.clip(r[..., d], 0, S[d] - 1) for d in range(3)]  # :118
        lin = (idx[0] * S[1] + idx[1]) * S[2] + idx[2]  # sub2ind :537-548
        return flat[lin]
    assert method == 'linear'
    loc0 = np.floor(loc)
    mx = [F(S[d] - 1) for d in range(3)]
    cl = [np.clip(loc[..., d], F(0), mx[d]) for d in range(3)]  # :72
    l0 = [np.clip(loc0[..., d], F(0), mx[d]) for d in range(3)]  # :73
    l1 = [np.clip(l0[d] + F(1), F(0), mx[d]) for d in range(3)]  # :76
    locs = [[a.astype(np.int32) for a in l0], [a.astype(np.int32) for a in l1]]
    d1 = [l1[d] - cl[d] for d in range(3)]  # :82
    d0 = [F(1) - d1[d] for d in range(3)]  # :83
    w = [d1, d0]  # :84
    acc = None
    for c in itertools.product([0, 1], repeat=3):  # :88
        lin = (locs[c[0]][0] * S[1] + locs[c[1]][1]) * S[2] + locs[c[2]][2]
        val = flat[lin].astype(F)
        wt = (w[c[0]][0] * w[c[1]][1]) * w[c[2]][2]  # prod_n :530-534
        term = wt[..., None] * val
        acc = term if acc is None else acc + term  # :110  (0 + x == x)
    return acc


def transform(vol, shift, method='linear'):
    """ext/neuron/utils.py:289-320: sample vol at mesh + shift"""
    shift = f32(shift)
    mesh = _grid(shift.shape[:-1])
    loc = np.stack([mesh[d] + shift[..., d] for d in range(3)], -1)
    return interpn(vol, loc, method)


def resize(vol, new_shape, method='linear'):
    """ext/neuron/utils.py:127-154 via nrn_layers.Resize (ext/neuron/layers.py:361-394):
    zoom = new/in (python float -> float32), sample position = i + (i/zoom - i)"""
    in_shape = vol.shape[:3]
    zoom = [F(new_shape[d] / in_shape[d]) for d in range(3)]
    grid = _grid(new_shape)
    offset = np.stack([grid[d] / zoom[d] - grid[d] for d in range(3)], -1)  # :150
    return transform(vol, offset, method)


def integrate_vec(vec, nb_steps=7):
    """scaling and squaring, ext/neuron/utils.py:365-369"""
    vec = f32(vec) / F(2 ** nb_steps)
    for _ in range(nb_steps):
        vec = vec + transform(vec, vec, 'linear')
    return vec


def matmul_f32(a, b):
    """k-ordered float32 multiply/add chain, no FMA (convention, see module docstring)"""
    a, b = f32(a), f32(b)
    acc = a[..., :, 0:1] * b[..., 0:1, :]
    for k in range(1, a.shape[-1]):
        acc = acc + a[..., :, k:k + 1] * b[..., k:k + 1, :]
    return acc


def affine_elastic_shift(aff, field, shape):
    """combine_non_linear_and_aff_to_shift / affine_to_shift (ext/neuron/utils.py:160-286).
    aff [4,4] float32; field [X,Y,Z,3] or None.  Returns shift [X,Y,Z,3]."""
    mesh = _grid(shape)
    mesh = [mesh[d] - F((shape[d] - 1) / 2) for d in range(3)]  # :271
    if field is None:
        flat = [m.reshape(-1) for m in mesh]
    else:
        field = f32(field)
        flat = [(mesh[d] + field[..., d]).reshape(-1) for d in range(3)]  # :276
    flat.append(np.ones_like(flat[0]))
    mesh_matrix = np.stack(flat, 0)  # 4 x N
    loc = matmul_f32(aff, mesh_matrix)[:3].T.reshape(list(shape) + [3])  # :281-283
    return loc - np.stack(mesh, -1)  # :286


# =============================================================================================
# affine sampling — ext/lab2im/utils.py:675-815
# =============================================================================================
def _uniform(u, lo, hi):
    """tf.random.uniform(minval, maxval): rnd * (max - min) + min in float32"""
    lo, hi = f32(lo), f32(hi)
    return f32(u) * (hi - lo) + lo


def _bounds(b, centre, size):
    """draw_value_from_distribution's hyperparameter handling (utils.py:1002-1016) for the cases the
    hot path uses: number -> [centre-b, centre+b]; (2,size) array -> rows."""
    if isinstance(b, np.ndarray):
        assert b.shape == (2, size)
        return b[0].astype(np.float64), b[1].astype(np.float64)
    if isinstance(b, (list, tuple)):
        assert len(b) == 2
        return np.full(size, b[0], np.float64), np.full(size, b[1], np.float64)
    return np.full(size, centre - b, np.float64), np.full(size, centre + b, np.float64)


def rotation_matrix(rot_deg):
    """create_rotation_transform, utils.py:755-782, float32; rot_deg [3] float32 degrees"""
    r = f32(rot_deg) * F(np.pi) / F(180)
    c, s = np.cos(r), np.sin(r)
    o, z = F(1), F(0)
    Rx = f32([[o, z, z], [z, c[0], -s[0]], [z, s[0], c[0]]])
    Ry = f32([[c[1], z, s[1]], [z, o, z], [-s[1], z, c[1]]])
    Rz = f32([[c[2], -s[2], z], [s[2], c[2], z], [z, z, o]])
    return matmul_f32(matmul_f32(Rx, Ry), Rz)


def sample_affine(u_rot=None, u_shear=None, u_scale=None, u_trans=None, rotation_bounds=False,
                  scaling_bounds=False, shearing_bounds=False, translation_bounds=False):
    """sample_affine_transform (utils.py:675-752) for one batch item from raw uniforms.
    Draw order on the tape: rot(3), shear(6), scale(3), trans(3), each only if its bound is not False."""
    eye = np.eye(3, dtype=F)
    if rotation_bounds is not False:
        lo, hi = _bounds(rotation_bounds, 0., 3)
        R = rotation_matrix(_uniform(u_rot, lo, hi))
    else:
        R = eye
    if shearing_bounds is not False:
        lo, hi = _bounds(shearing_bounds, 0., 6)
        h = _uniform(u_shear, lo, hi)
        Sh = f32([[1, h[0], h[1]], [h[2], 1, h[3]], [h[4], h[5], 1]])  # :801-807
    else:
        Sh = eye
    if scaling_bounds is not False:
        lo, hi = _bounds(scaling_bounds, 1., 3)
        S = np.diag(_uniform(u_scale, lo, hi)).astype(F)
    else:
        S = eye
    T = matmul_f32(S, matmul_f32(Sh, R))  # :735
    aff = np.zeros((4, 4), F)
    aff[:3, :3] = T
    if translation_bounds is not False:
        lo, hi = _bounds(translation_bounds, 0., 3)
        aff[:3, 3] = _uniform(u_trans, lo, hi)
    aff[3, 3] = 1
    return aff


# =============================================================================================
# lab2im layers — ext/lab2im/layers.py, ext/lab2im/edit_tensors.py
# =============================================================================================
def get_resample_shape(shape, factor):
    """utils.py:577-588"""
    return [math.ceil(shape[i] * factor) for i in range(len(shape))]


def random_spatial_deformation(vols, methods, aff, u_std, n_field, nonlin_std, nonlin_scale):
    """RandomSpatialDeformation.call (layers.py:161-211) for one batch item.
    vols: list of [X,Y,Z,C]; aff [4,4] or None; u_std raw uniform scalar, n_field raw normals [*small,3]."""
    shape = vols[0].shape[:3]
    field = None
    if nonlin_std > 0:
        small = get_resample_shape(shape, nonlin_scale)
        std = _uniform(u_std, 0., nonlin_std)  # :189
        svf = f32(n_field).reshape(small + [3]) * std  # :190 (rnd*stddev + 0)
        half = [max(int(shape[i] / 2), small[i]) for i in range(3)]  # :193
        svf = resize(svf, half, 'linear')
        svf = integrate_vec(svf, 7)
        field = resize(svf, list(shape), 'linear')
    if aff is None and field is None:
        return [np.asarray(v) for v in vols], None
    # a single dense transform is used as the shift directly (ext/neuron/layers.py:148-151)
    shift = field if aff is None else affine_elastic_shift(aff, field, shape)
    outs = []
    for v, m in zip(vols, methods):
        o = transform(f32(v), shift, m)  # inputs are cast to float32 (:167) ...
        outs.append(o.astype(v.dtype))  # ... and back (:209-211)
    return outs, shift


def random_crop_index(u_crop, in_shape, crop_shape):
    """RandomCrop._single_slice (layers.py:266-270): int32(U * (in - crop))"""
    mx = f32(np.array(in_shape) - np.array(crop_shape))
    return (f32(u_crop) * (mx - F(0)) + F(0)).astype(np.int32)


def swap_lut(label_list, n_neutral):
    """RandomFlip.build (layers.py:375-386) + utils.get_mapping_lut (utils.py:894-914); None if no sided labels"""
    label_list = np.asarray(label_list)
    n = len(label_list)
    if n_neutral == n:
        return None
    half = n_neutral + int((n - n_neutral) / 2)
    swapped = np.concatenate([label_list[:n_neutral], label_list[half:], label_list[n_neutral:half]])
    lut = np.zeros(np.max(label_list) + 1, dtype=np.int32)
    for s, d in zip(label_list, swapped):
        lut[s] = d
    return lut


def random_flip(vols, u_flip, lut=None, swap=(True,), prob=0.5):
    """RandomFlip.call (layers.py:391-427) as SynthSR uses it (flip_axes=[0])"""
    do = bool(f32(u_flip) < F(prob))
    outs = []
    for v, s in zip(vols, swap):
        if do and s and lut is not None:
            v = lut[v]
        if do:
            v = v[::-1]
        outs.append(np.ascontiguousarray(v))
    return outs, do


def gmm_lut(generation_labels, values):
    """SampleConditionalGMM's scatter_nd LUT for ONE batch item (layers.py:472-490).
    values [n_labels, n_channels] -> lut [n_channels, max_label+1]"""
    generation_labels = np.asarray(generation_labels)
    values = f32(values)
    lut = np.zeros((values.shape[1], int(generation_labels.max()) + 1), F)
    for c in range(values.shape[1]):
        np.add.at(lut[c], generation_labels, values[:, c])
    return lut


def sample_gmm(labels, generation_labels, means, stds, noise):
    """SampleConditionalGMM.call (layers.py:480-498): labels int [X,Y,Z], means/stds [L,C], noise [X,Y,Z,C]"""
    lm, ls = gmm_lut(generation_labels, means), gmm_lut(generation_labels, stds)
    C = lm.shape[0]
    mm = np.stack([lm[c][labels] for c in range(C)], -1)
    sm = np.stack([ls[c][labels] for c in range(C)], -1)
    return sm * f32(noise) + mm  # :498


def sample_gmm_batch(labels, generation_labels, means, stds, noise):
    """SampleConditionalGMM.call on a batch (layers.py:480-498) AS THE REFERENCE COMPUTES IT: the scatter indices are tiled
    over the batch (:482-483) and scattered into one table of shape [max index + 1] (:490, :495), where tf.scatter_nd adds up
    what lands on the same index -- the per-label means / stds of all B items are SUMMED, the table is tiled back over the
    batch and every item gathers from the sums (pinned by tests/golden/gmm_batch.npz).
    labels int [B,X,Y,Z], means / stds [B,L,C], noise [B,X,Y,Z,C]"""
    msum, ssum = f32(means).sum(0, dtype=F), f32(stds).sum(0, dtype=F)
    return np.stack([sample_gmm(labels[b], generation_labels, msum, ssum, noise[b]) for b in range(len(labels))], 0)


def bias_field(x, u_std, n_small, u_gate, bias_std, bias_scale, prob=0.95):
    """BiasFieldCorruption.call (layers.py:1067-1097) for one single-channel volume x [X,Y,Z,1]"""
    shape = x.shape[:3]
    small = get_resample_shape(shape, bias_scale)
    std = _uniform(u_std, 0., bias_std)  # :1080
    b = f32(n_small).reshape(small + [1]) * std
    b = np.exp(resize(b, list(shape), 'linear'))  # :1083-1084
    if bool(f32(u_gate) < F(prob)):  # :1090
        return b * f32(x), True
    return f32(x), False


def intensity_augmentation(x, n_gamma=None, clip=300, normalise=True, gamma_std=0.5):
    """IntensityAugmentation.call (layers.py:1186-1257), separate_channels, no noise, one channel [X,Y,Z,1]"""
    x = f32(x)
    if clip:
        x = np.clip(x, F(0), F(clip))  # :1215
    if normalise:
        m, M = x.min(), x.max()  # :1230-1231
        x = np.clip(x, m, M)
        x = (x - m) / (M - m + F(1e-7))  # :1236
    if gamma_std > 0:
        g = f32(n_gamma) * F(gamma_std)  # :1240
        x = np.power(x, np.exp(g))  # :1242
    return x


def blur_window(sigma):
    """edit_tensors.py:124"""
    return (np.int32(np.ceil(2.5 * np.array(sigma, dtype=np.float64)) / 2) * 2 + 1).tolist()


def gaussian_kernel(sigma, u_blur=None, blur_range=None):
    """et.gaussian_kernel, non-separable branch (edit_tensors.py:86-181), float32.
    sigma: 3 floats; u_blur: raw uniform [3] (only if blur_range not None and != 1)."""
    sig = f32(sigma)
    if blur_range is not None and blur_range != 1:
        sig = sig * _uniform(u_blur, 1 / blur_range, blur_range)  # :121
    ws = blur_window(sigma)
    mesh = _grid(ws)
    diff = np.stack([mesh[d] - F((ws[d] - 1) / 2) for d in range(3)], -1)
    s = sig.reshape(1, 1, 1, 3)
    is0 = s == 0
    s1 = np.where(is0, F(1), s)
    exp_term = -np.square(diff) / (F(2) * s1 ** 2)  # :174
    norms = exp_term - np.log(np.where(is0, F(1), F(np.sqrt(2 * np.pi)) * s))  # :175
    k = np.exp(np.sum(norms, -1, dtype=F))  # :176-177
    return k / np.sum(k, dtype=F)  # :178


def conv3d_same(x, k):
    """tf.nn.conv3d(..., 'SAME') for one channel: zero-padded cross-correlation (layers.py:758).
    float32 accumulation in (z,y,x) raster order of the taps.  x [X,Y,Z], k [a,b,c]"""
    x, k = f32(x), f32(k)
    pz, py, px = k.shape[0] // 2, k.shape[1] // 2, k.shape[2] // 2
    xp = np.pad(x, ((pz, pz), (py, py), (px, px)))
    Z, Y, X = x.shape
    acc = np.zeros_like(x)
    for a in range(k.shape[0]):
        for b in range(k.shape[1]):
            for c in range(k.shape[2]):
                acc = acc + xp[a:a + Z, b:b + Y, c:c + X] * k[a, b, c]
    return acc


def gaussian_blur(x, sigma, u_blur=None, blur_range=None):
    """GaussianBlur.call (layers.py:732-767); x [X,Y,Z,1].  |sigma| > 5 (layers.py:720): separable branch = one 1-D kernel
    per axis whose window is wider than 1, applied in turn (:747-749) -- the same kernels as dynamic_gaussian_blur with
    max_sigma = sigma"""
    if np.linalg.norm(np.array(sigma, dtype=np.float64)) > 5:
        return dynamic_gaussian_blur(x, sigma, u_blur, sigma, blur_range)
    if not any(sigma):
        return f32(x)
    k = gaussian_kernel(sigma, u_blur, blur_range)
    return conv3d_same(x[..., 0], k)[..., None]


def sample_resolution(u_axis, u_res, u_gate, u_thick, min_res, max_res_iso, prob_min=0.05):
    """SampleResolution.call (ext/lab2im/layers.py:598-652) as SynthSR builds it (`SampleResolution(atlas_res, max_res)`:
    max_res_iso only, return_thickness=True), one batch item.  Raw draws in call order: u_axis (the int axis pick, unused
    by this branch but drawn, :609), u_res [3] (:625 -- three independent values despite the name), u_gate (:626),
    u_thick [3] (:649).  Returns (resolution [3], thickness [3]) float32."""
    del u_axis
    lo, hi = f32(min_res), f32(max_res_iso)
    res = f32(u_res) * (hi - lo) + lo
    if F(f32(u_gate).reshape(-1)[0]) < F(prob_min):
        res = lo.copy()
    thick = f32(u_thick) * (res - lo) + lo
    return res, thick


def dynamic_gaussian_blur(x, sigma, u_blur, max_sigma, blur_range):
    """DynamicGaussianBlur.call, separable branch (layers.py:810-817, |max_sigma| > 5) with
    edit_tensors.gaussian_kernel(sigma tensor, max_sigma, blur_range, separable=True) (:86-154): per axis a 1-D kernel
    of ceil(2.5 max_sigma)-derived width, exp(-d^2/2s^2 - log(sqrt(2 pi) s)) normalised to sum 1, applied one axis
    after the other with zero padding.  x [X,Y,Z,1]; sigma, u_blur [3]."""
    sig = f32(sigma)
    if blur_range is not None and blur_range != 1:
        sig = sig * (f32(u_blur) * (F(blur_range) - F(1 / blur_range)) + F(1 / blur_range))
    win = blur_window(max_sigma)
    out = f32(x)[..., 0]
    for a in range(3):
        if win[a] > 1:
            loc = np.arange(win[a]).astype(F) - F((win[a] - 1) / 2)
            e = -np.square(loc) / (F(2) * sig[a] ** 2)
            g = np.exp(e - np.log(F(np.sqrt(2 * np.pi)) * sig[a]))
            g = (g / g.sum(dtype=F)).astype(F)
            shape = [1, 1, 1]
            shape[a] = win[a]
            out = conv3d_same(out, g.reshape(shape))
    return out[..., None]


def mimic_acquisition(x, subsample_res, volume_res, resample_shape):
    """MimicAcquisition.call (layers.py:927-990) with min_subsample_res = volume_res, no noise, build_dist_map=True:
    nearest-neighbour down-sampling to int(shape*volume_res/subsample_res) voxels -- kept inside a full-size tensor, so
    entries beyond the down-sampled extent are edge replicas (the sampling positions are clipped, :953) -- then linear
    up-sampling to resample_shape, plus the distance (mm) of every output voxel to the nearest acquired grid point.
    x [X,Y,Z,1]; returns (volume [*resample_shape,1], dist [*resample_shape,1])."""
    x = f32(x)
    S = np.array(x.shape[:3])
    sub = f32(subsample_res)
    down_shape = (f32(S * np.array(volume_res, dtype=np.float64)) / sub).astype(np.int32)  # :943-944
    down_zoom = (down_shape / S).astype(F)  # :945 (true division of ints, cast to float32)
    up_zoom = (np.array(resample_shape, dtype=np.int32) / down_shape).astype(F)  # :946
    g = np.stack(_grid(list(S)), -1)
    down_loc = np.clip(g / down_zoom, F(0), f32(S))  # :949-953
    vol = interpn(x, down_loc, 'nearest')
    gu = np.stack(_grid(list(resample_shape)), -1)
    up_loc = f32(gu / up_zoom)  # :968-969
    vol = interpn(vol, up_loc, 'linear')
    fl, ce = np.floor(up_loc), np.ceil(up_loc)
    dist = np.minimum(up_loc - fl, ce - up_loc) * sub  # :984-986
    dist = np.sqrt(np.sum(np.square(dist), -1, keepdims=True, dtype=F)).astype(F)
    return vol, dist


def blurring_sigma_for_downsampling(current_res, downsample_res, mult_coef=None, thickness=None):
    """edit_tensors.py:41-65 (numpy branch, float64 like the reference)"""
    current_res = np.array(current_res, dtype=np.float64)
    downsample_res = np.array(downsample_res, dtype=np.float64)
    if thickness is not None:
        downsample_res = np.minimum(downsample_res, np.array(thickness, dtype=np.float64))
    if mult_coef is None:
        sigma = 0.75 * downsample_res / current_res
        sigma[downsample_res == current_res] = 0.5
    else:
        sigma = mult_coef * downsample_res / current_res
    sigma[downsample_res == 0] = 0
    return sigma


def reliability_map_1d(n_out, n_down):
    """edit_tensors.py:313-323 for one axis"""
    up = n_out / n_down
    loc = np.arange(0, n_out, up)
    fl = np.int32(np.floor(loc))
    ce = np.int32(np.clip(fl + 1, 0, n_out - 1))
    w = np.zeros(n_out)
    w[fl] = 1 - (loc - fl)
    w[ce] = w[ce] + (loc - fl)
    return w


def resample_tensor(x, resample_shape, subsample_res=None, volume_res=None, build_reliability_map=False):
    """et.resample_tensor (edit_tensors.py:257-338), interp 'linear'; x [X,Y,Z,1]"""
    shape = list(x.shape[:3])
    down = shape
    if subsample_res is not None and list(subsample_res) != list(volume_res):
        down = [int(shape[i] * volume_res[i] / subsample_res[i]) for i in range(3)]  # :295
        x = resize(x, down, 'nearest')
    if list(resample_shape) != down:
        x = resize(x, list(resample_shape), 'linear')
    if not build_reliability_map:
        return x, None
    if down != shape:
        rel = 1
        for i in range(3):
            sh = [1, 1, 1]
            sh[i] = resample_shape[i]
            rel = rel * reliability_map_1d(resample_shape[i], down[i]).reshape(sh)  # float64 product :326
        rel = f32(rel)[..., None]
    else:
        rel = np.ones_like(x)
    return x, rel


def find_closest_number_divisible_by_m(n, m):
    """utils.py:928-944 ('lower')"""
    return n if n % m == 0 else int(n / m) * m


def get_shapes(labels_shape, output_shape, atlas_res, target_res, padding_margin, output_div_by_n):
    """SynthSR/labels_to_image_model.py:269-335 -> (cropping_shape, output_shape)"""
    n = 3
    atlas_res, target_res = list(atlas_res), list(target_res)
    labels_shape = list(labels_shape)
    if padding_margin is not None:
        pm = [padding_margin] * n if np.isscalar(padding_margin) else list(padding_margin)
        labels_shape = [labels_shape[i] + 2 * int(pm[i]) for i in range(n)]
    factor = [atlas_res[i] / float(target_res[i]) for i in range(n)] if atlas_res != target_res else None
    if output_shape is not None:
        output_shape = [int(output_shape)] * n if np.isscalar(output_shape) else [int(s) for s in output_shape]
        if factor is not None:
            output_shape = [min(int(labels_shape[i] * factor[i]), output_shape[i]) for i in range(n)]
        else:
            output_shape = [min(labels_shape[i], output_shape[i]) for i in range(n)]
        if output_div_by_n is not None:
            output_shape = [find_closest_number_divisible_by_m(s, output_div_by_n) for s in output_shape]
        if factor is not None:
            crop = [int(np.around(output_shape[i] / factor[i], 0)) for i in range(n)]
        else:
            crop = output_shape
    else:
        if output_div_by_n is not None:
            if factor is not None:
                output_shape = [int(labels_shape[i] * factor[i]) for i in range(n)]
                output_shape = [find_closest_number_divisible_by_m(s, output_div_by_n) for s in output_shape]
                crop = [int(np.around(output_shape[i] / factor[i], 0)) for i in range(n)]
            else:
                crop = [find_closest_number_divisible_by_m(s, output_div_by_n) for s in labels_shape]
                output_shape = crop
        else:
            crop = labels_shape
            output_shape = [int(crop[i] * factor[i]) for i in range(n)] if factor is not None else crop
    return crop, output_shape


# =============================================================================================
# whole graph — SynthSR/labels_to_image_model.py:32-266, batch 1, synthetic target
# =============================================================================================
class TapeReader:
    """serves tape entries in order, checking kind ('u'/'n') and element count"""

    def __init__(self, entries):
        self.entries = list(entries)
        self.pos = 0

    def get(self, kind, size):
        k, a = self.entries[self.pos]
        a = np.asarray(a)
        assert k == kind and a.size == size, (self.pos, k, kind, a.shape, size)
        self.pos += 1
        return f32(a).reshape(-1)

    def done(self):
        return self.pos == len(self.entries)


def labels_to_image(labels, means, stds, tape, generation_labels, n_neutral_labels, input_channels,
                    output_channel, atlas_res=(1., 1., 1.), target_res=None, output_shape=None,
                    output_div_by_n=None, padding_margin=None, flipping=True, scaling_bounds=.15,
                    rotation_bounds=15, shearing_bounds=.012, translation_bounds=False, nonlin_std=3.,
                    nonlin_shape_factor=.0625, simulate_registration_error=True, data_res=None, thickness=None,
                    downsample=False, build_reliability_maps=False, blur_range=1.15, bias_field_std=.3,
                    bias_shape_factor=.025, real_image=None, randomise_res=False):
    """labels int32 [X,Y,Z]; means/stds [L,C]; tape: list of (kind, array) in the reference's call order.
    output_channel=None selects the real-image regression target (labels_to_image_model.py:71,109-113): `real_image`
    float32 [X,Y,Z] is deformed (linear), cropped and flipped jointly with the labels, then min-max normalised (:248-255).
    Returns dict(image [X',Y',Z',Cin(+maps)], target [X',Y',Z',Ct], seg int32 [X',Y',Z'], extras)."""
    tp = tape if isinstance(tape, TapeReader) else TapeReader(tape)
    input_channels = list(input_channels)
    n_channels = len(input_channels)
    idx_first = int(np.argmax(input_channels))
    if not isinstance(simulate_registration_error, (list, tuple)):
        simulate_registration_error = [simulate_registration_error] * n_channels
    atlas = np.tile(np.array(atlas_res, dtype=np.float64).reshape(1, 3), (n_channels, 1))
    if data_res is not None:
        data_res = np.array(data_res, dtype=np.float64)
        thickness = None if thickness is None else np.array(thickness, dtype=np.float64)
    if output_channel is not None:
        for idx in output_channel:  # :86-89
            if not input_channels[idx]:
                data_res = np.insert(data_res, idx, 1, axis=0) if data_res is not None else None
                thickness = np.insert(thickness, idx, 1, axis=0) if thickness is not None else None
    data_res = atlas if data_res is None else np.broadcast_to(np.atleast_2d(data_res), (n_channels, 3)).copy()
    thickness = data_res if thickness is None else np.broadcast_to(np.atleast_2d(thickness), (n_channels, 3)).copy()
    down = [bool(downsample)] * n_channels if downsample else list(np.min(thickness - data_res, 1) < 0)
    atlas_res = list(atlas[0])
    target_res = atlas_res if target_res is None else list(np.array(target_res, dtype=np.float64).reshape(-1)[:3])
    labels = np.asarray(labels)
    assert labels.ndim == 3
    use_real = output_channel is None
    if use_real:
        assert real_image is not None and tuple(np.shape(real_image)) == labels.shape, 'real image of the label-map shape'
        real = f32(real_image)
    crop_shape, out_shape = get_shapes(labels.shape, output_shape, atlas_res, target_res, padding_margin,
                                       output_div_by_n)
    if padding_margin is not None:  # PadAroundCentre, layers.py:1754
        pm = [int(padding_margin)] * 3 if np.isscalar(padding_margin) else [int(p) for p in padding_margin]
        labels = np.pad(labels, [(p, p) for p in pm])
        if use_real:
            real = np.pad(real, [(p, p) for p in pm])  # :119-120
    lshape = list(labels.shape)

    # --- deformation (:124-142)
    any_aff = any(b is not False for b in (scaling_bounds, rotation_bounds, shearing_bounds, translation_bounds))
    aff = None
    if any_aff:
        aff = sample_affine(tp.get('u', 3) if rotation_bounds is not False else None,
                            tp.get('u', 6) if shearing_bounds is not False else None,
                            tp.get('u', 3) if scaling_bounds is not False else None,
                            tp.get('u', 3) if translation_bounds is not False else None,
                            rotation_bounds, scaling_bounds, shearing_bounds, translation_bounds)
    u_std = n_field = None
    if nonlin_std > 0:
        small = get_resample_shape(lshape, nonlin_shape_factor)
        u_std = tp.get('u', 1)[0]
        n_field = tp.get('n', int(np.prod(small)) * 3)
    if use_real:  # inter_method=['nearest', 'linear'] (:128-134)
        (lab, real), shift = random_spatial_deformation([labels.astype(np.int32)[..., None], real[..., None]],
                                                        ['nearest', 'linear'], aff, u_std, n_field, nonlin_std,
                                                        nonlin_shape_factor)
        real = real[..., 0]
    else:
        (lab,), shift = random_spatial_deformation([labels.astype(np.int32)[..., None]], ['nearest'], aff, u_std,
                                                   n_field, nonlin_std, nonlin_shape_factor)
    lab = lab[..., 0]
    extras = dict(affine=aff, shift=shift)
    # --- crop (:145-151)
    if crop_shape != lshape:
        ci = random_crop_index(tp.get('u', 3), lshape, crop_shape)
        lab = lab[ci[0]:ci[0] + crop_shape[0], ci[1]:ci[1] + crop_shape[1], ci[2]:ci[2] + crop_shape[2]]
        if use_real:  # the same crop (RandomCrop on the concatenated inputs, layers.py:252-270)
            real = real[ci[0]:ci[0] + crop_shape[0], ci[1]:ci[1] + crop_shape[1], ci[2]:ci[2] + crop_shape[2]]
        extras['crop_idx'] = ci
    # --- flip (:154-162)
    if flipping:
        if use_real:  # swap_labels=True only applies to the label map (layers.py:382-386)
            (lab, real), flipped = random_flip([lab, real], tp.get('u', 1)[0],
                                               swap_lut(generation_labels, n_neutral_labels), swap=(True, False))
        else:
            (lab,), flipped = random_flip([lab], tp.get('u', 1)[0], swap_lut(generation_labels, n_neutral_labels))
        extras['flipped'] = flipped
    # --- GMM (:166)
    S = lab.shape
    nvox = int(np.prod(S))
    image = sample_gmm(lab, generation_labels, means, stds, tp.get('n', nvox * n_channels).reshape(*S, n_channels))
    channels, targets = [], []
    for i in range(n_channels):
        ch = image[..., i:i + 1]
        if input_channels[i]:  # :178-180
            small_b = get_resample_shape(list(S), bias_shape_factor)
            if bias_field_std > 0:
                ch, _ = bias_field(ch, tp.get('u', 1)[0], tp.get('n', int(np.prod(small_b))), tp.get('u', 1)[0],
                                   bias_field_std, bias_shape_factor)
        ch = intensity_augmentation(ch, tp.get('n', 1)[0], clip=300, normalise=True, gamma_std=.5)  # :184
        ch = gaussian_blur(ch, [.5] * 3)  # :186
        if output_channel is not None and any(c == i for c in output_channel):  # :189-196
            tgt = ch
            if crop_shape != out_shape:
                sig = blurring_sigma_for_downsampling(atlas_res, target_res)
                tgt = gaussian_blur(tgt, list(sig))
                tgt, _ = resample_tensor(tgt, out_shape)
            targets.append(tgt)
        if input_channels[i]:
            reg = bool(simulate_registration_error[i]) and (i != idx_first)
            Tinv = None
            if reg:  # :202-208
                T = sample_affine(tp.get('u', 3), None, None, tp.get('u', 3), rotation_bounds=5,
                                  translation_bounds=5)
                Tinv = np.linalg.inv(T.astype(np.float64)).astype(F)
                ch = transform(ch, affine_elastic_shift(T, None, S), 'linear')
            rr = randomise_res[i] if isinstance(randomise_res, (list, tuple)) else randomise_res
            if rr:  # :215-220: random acquisition resolution, separable blur, acquisition mimicking + distance map
                max_res = [9.] * 3
                res, thick = sample_resolution(tp.get('u', 1), tp.get('u', 3), tp.get('u', 1), tp.get('u', 3), atlas_res,
                                               max_res)
                # blurring_sigma_for_downsampling, tensor branch (edit_tensors.py:67-82), float32
                dres = np.minimum(res, thick)
                sig = np.where(dres == 0, F(0), F(.42) * dres / f32(atlas_res))
                ch = dynamic_gaussian_blur(ch, sig, tp.get('u', 3) if (blur_range is not None and blur_range != 1)
                                           else None, 0.75 * np.array(max_res) / np.array(atlas_res), blur_range)
                ch, rel = mimic_acquisition(ch, res, atlas_res, out_shape)
            else:
                sig = blurring_sigma_for_downsampling(atlas_res, data_res[i], .42, thickness[i])  # :223
                ch = gaussian_blur(ch, list(sig), tp.get('u', 3) if (blur_range is not None and blur_range != 1)
                                   else None, blur_range)
                if down[i]:
                    ch, rel = resample_tensor(ch, out_shape, list(data_res[i]), atlas_res, True)  # :226
                else:
                    ch, rel = resample_tensor(ch, out_shape, build_reliability_map=True)  # :228
            if reg:  # :231-238
                Terr = sample_affine(tp.get('u', 3), None, None, tp.get('u', 3), rotation_bounds=.5,
                                     translation_bounds=.5)
                Tie = matmul_f32(Terr, Tinv)
                sh = affine_elastic_shift(Tie, None, ch.shape[:3])
                ch = transform(ch, sh, 'linear')
                rel = transform(rel, sh, 'linear')
            channels.append(ch)
            if build_reliability_maps:
                channels.append(rel)
    if use_real:  # :248-255
        tgt = intensity_augmentation(real[..., None], None, clip=False, normalise=True, gamma_std=0)
        if crop_shape != out_shape:
            tgt = gaussian_blur(tgt, list(blurring_sigma_for_downsampling(atlas_res, target_res)))
            tgt, _ = resample_tensor(tgt, out_shape)
        targets = [tgt]
    assert tp.done(), 'tape not fully consumed'
    return dict(image=np.concatenate(channels, -1), target=np.concatenate(targets, -1),
                seg=lab.astype(np.int32), extras=extras)
