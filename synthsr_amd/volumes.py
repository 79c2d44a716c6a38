"""Volume IO / orientation helpers the hot path's callers need — counterparts of
ext/lab2im/utils.py:76-160 (`load_volume`, `save_volume`), :163-206 (`get_volume_info`), :209-285
(`get_list_labels`), :391-424 (`list_images_in_folder`) and ext/lab2im/edit_volumes.py:591-654
(`get_ras_axes`, `align_volume_to_ref`) and :504-556 (`resample_volume`).  NIfTI only (own reader, nifti.py; nibabel is not installed).
"""
import glob
import os
import numpy as np

from . import host_math as hm
from .nifti import read_nifti, write_nifti

get_ras_axes = hm.get_ras_axes


def get_dims(shape, max_channels=10):
    """ext/lab2im/utils.py:558-574"""
    if shape[-1] <= max_channels:
        return len(shape) - 1, shape[-1]
    return len(shape), 1


def list_images_in_folder(path_dir, include_single_image=True, check_if_empty=True):
    basename = os.path.basename(path_dir)
    if include_single_image and (('.nii.gz' in basename) or ('.nii' in basename) or ('.mgz' in basename) or
                                 ('.npz' in basename)):
        assert os.path.isfile(path_dir), 'file %s does not exist' % path_dir
        return [path_dir]
    if not os.path.isdir(path_dir):
        raise Exception('Folder does not exist: %s' % path_dir)
    lst = sorted(glob.glob(os.path.join(path_dir, '*nii.gz')) + glob.glob(os.path.join(path_dir, '*nii')) +
                 glob.glob(os.path.join(path_dir, '*.mgz')) + glob.glob(os.path.join(path_dir, '*.npz')))
    if check_if_empty:
        assert len(lst) > 0, 'no .nii, .nii.gz, .mgz or .npz image could be found in %s' % path_dir
    return lst


def align_volume_to_ref(volume, aff, aff_ref=None, return_aff=False, n_dims=None, return_copy=True):
    """ext/lab2im/edit_volumes.py:609-654"""
    new_volume = volume.copy() if return_copy else volume
    aff_flo = np.array(aff, dtype=np.float64)
    if aff_ref is None:
        aff_ref = np.eye(4)
    if n_dims is None:
        n_dims, _ = get_dims(new_volume.shape)
    ras_ref = get_ras_axes(aff_ref, n_dims=n_dims)
    ras_flo = get_ras_axes(aff_flo, n_dims=n_dims)
    aff_flo[:, ras_ref] = aff_flo[:, ras_flo]
    for i in range(n_dims):
        if ras_flo[i] != ras_ref[i]:
            new_volume = np.swapaxes(new_volume, ras_flo[i], ras_ref[i])
            swapped = int(np.where(ras_flo == ras_ref[i])[0][0])
            ras_flo[swapped], ras_flo[i] = ras_flo[i], ras_flo[swapped]
    dots = np.sum(aff_flo[:3, :3] * np.asarray(aff_ref)[:3, :3], axis=0)
    for i in range(n_dims):
        if dots[i] < 0:
            new_volume = np.flip(new_volume, axis=i)
            aff_flo[:, i] = -aff_flo[:, i]
            aff_flo[:3, 3] = aff_flo[:3, 3] - aff_flo[:3, i] * (new_volume.shape[i] - 1)
    if return_aff:
        return new_volume, aff_flo
    return new_volume


def load_volume(path_volume, im_only=True, squeeze=True, dtype=None, aff_ref=None):
    assert path_volume.endswith(('.nii', '.nii.gz', '.mgz', '.npz')), 'Unknown data file: %s' % path_volume
    if path_volume.endswith(('.nii', '.nii.gz', '.mgz')):
        if path_volume.endswith('.mgz'):
            from .mgh import read_mgh
            data, aff, header = read_mgh(path_volume)
        else:
            data, aff, header = read_nifti(path_volume)
        volume = np.asarray(data, dtype=np.float64)  # nibabel get_fdata() semantics
        if squeeze:
            volume = np.squeeze(volume)
    else:
        volume = np.load(path_volume)['vol_data']
        if squeeze:
            volume = np.squeeze(volume)
        aff, header = np.eye(4), dict(pixdim=(1.,) * 8)
    if dtype is not None:
        if 'int' in dtype:
            volume = np.round(volume)
        volume = volume.astype(dtype={'int': np.int64}.get(dtype, dtype))
    if aff_ref is not None:
        n_dims, _ = get_dims(list(volume.shape), max_channels=10)
        volume, aff = align_volume_to_ref(volume, aff, aff_ref=aff_ref, return_aff=True, n_dims=n_dims)
    if im_only:
        return volume
    return volume, aff, header


def save_volume(volume, aff, header, path, res=None, dtype=None, n_dims=3):
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    if '.npz' in path:
        np.savez_compressed(path, vol_data=volume)
        return
    if isinstance(aff, str) and aff == 'FS':
        aff = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1]])
    elif aff is None:
        aff = np.eye(4)
    if dtype is not None and 'int' in dtype:
        volume = np.round(volume)
    if path.endswith(('.mgz', '.mgh')):
        from .mgh import write_mgh
        write_mgh(path, volume, aff, dtype=dtype)
        return
    write_nifti(path, volume, aff, dtype=dtype)


def _axis_samples(n, factor):
    """sample positions (clamped to the grid) of one axis for a zoom `factor`: the reference's
    `arange(start, start + step*ceil(n*factor), step)` with start = -(f-1)/(2f), step = 1/f
    (ext/lab2im/edit_volumes.py:531-544)"""
    start = -(factor - 1.0) / (2.0 * factor)
    step = 1.0 / factor
    pos = np.arange(start=start, stop=start + step * np.ceil(n * factor), step=step)
    return np.clip(pos, 0.0, n - 1.0)


def _interp_axis(vol, pos, axis, nearest):
    """1-D interpolation of `vol` along `axis` at fractional positions `pos` (already inside [0, n-1])"""
    n = vol.shape[axis]
    i0 = np.minimum(np.floor(pos).astype(np.int64), max(n - 2, 0))
    w = pos - i0
    i1 = np.minimum(i0 + 1, n - 1)
    shape = [1] * vol.ndim
    shape[axis] = -1
    if nearest:  # scipy RegularGridInterpolator('nearest'): the upper neighbour only when strictly closer
        return np.take(vol, np.where(w <= 0.5, i0, i1), axis=axis)
    a, b = np.take(vol, i0, axis=axis), np.take(vol, i1, axis=axis)
    w = w.reshape(shape)
    return a * (1.0 - w) + b * w


def resample_volume(volume, aff, new_vox_size, interpolation='linear', blur=True):
    """Resize the voxels of a 3-D volume to `new_vox_size` (mm) and update the affine so that world coordinates are kept
    (ext/lab2im/edit_volumes.py:504-556, used by scripts/predict_command_line.py:114).  Down-sampled axes are first
    smoothed with a Gaussian of sigma = 0.25 / zoom; the separable (tri)linear resampling below is what the reference
    obtains from RegularGridInterpolator on its clamped sampling grid.  Returns (new volume, new affine)."""
    from scipy.ndimage import gaussian_filter
    if interpolation not in ('linear', 'nearest'):
        raise ValueError("interpolation should be 'linear' or 'nearest', had %s" % interpolation)
    volume = np.asarray(volume)
    aff = np.asarray(aff, dtype=np.float64)
    vox = np.sqrt(np.sum(aff * aff, axis=0))[:-1]
    factor = vox / np.array(new_vox_size, dtype=np.float64)
    sigmas = np.where(factor > 1, 0.0, 0.25 / factor)  # no anti-aliasing blur on up-sampled axes
    out = gaussian_filter(volume, sigmas) if blur else volume
    for ax in range(3):
        out = _interp_axis(out, _axis_samples(volume.shape[ax], factor[ax]), ax, interpolation == 'nearest')
    aff2 = aff.copy()
    aff2[:3, :3] = aff2[:3, :3] / factor[None, :]
    aff2[:3, 3] = aff2[:3, 3] - aff2[:3, :3] @ (0.5 * (factor - 1.0))
    return out, aff2


def resample_volume_like(vol_ref, aff_ref, vol_flo, aff_flo, interpolation='linear'):
    """Reslice the floating volume onto the grid of the reference volume (ext/lab2im/edit_volumes.py:558-590, used by
    scripts/predict_command_line_hyperfine.py:112): every reference voxel is mapped to floating voxel coordinates with
    inv(aff_flo) @ aff_ref and sampled (tri)linearly or nearest; positions outside the floating grid give 0."""
    if interpolation not in ('linear', 'nearest'):
        raise ValueError("interpolation should be 'linear' or 'nearest', had %s" % interpolation)
    vol_flo = np.asarray(vol_flo, dtype=np.float64)
    T = np.linalg.inv(np.asarray(aff_flo, dtype=np.float64)) @ np.asarray(aff_ref, dtype=np.float64)
    shape = tuple(np.shape(vol_ref)[:3])
    g = np.stack([a.reshape(-1) for a in np.meshgrid(*[np.arange(n) for n in shape], indexing='ij')] +
                 [np.ones(int(np.prod(shape)))])
    pos = (T @ g)[:3]
    n = np.array(vol_flo.shape).reshape(3, 1)
    inside = np.all((pos >= 0) & (pos <= n - 1), axis=0)
    pos = np.clip(pos, 0, n - 1)
    i0 = np.minimum(np.floor(pos).astype(np.int64), np.maximum(n - 2, 0))
    w = pos - i0
    i1 = np.minimum(i0 + 1, n - 1)
    if interpolation == 'nearest':
        idx = np.where(w <= 0.5, i0, i1)
        out = vol_flo[idx[0], idx[1], idx[2]]
    else:
        out = np.zeros(pos.shape[1])
        for cz in (0, 1):
            for cy in (0, 1):
                for cx in (0, 1):
                    wz = w[0] if cz else 1.0 - w[0]
                    wy = w[1] if cy else 1.0 - w[1]
                    wx = w[2] if cx else 1.0 - w[2]
                    out += wz * wy * wx * vol_flo[(i1 if cz else i0)[0], (i1 if cy else i0)[1], (i1 if cx else i0)[2]]
    out[~inside] = 0.0
    return out.reshape(shape)


def rescale_volume(volume, new_min=0, new_max=255, min_percentile=2, max_percentile=98, use_positive_only=False):
    """ext/lab2im/edit_volumes.py:148-176: clip to the [min_percentile, max_percentile] intensities (of the positive
    voxels only if use_positive_only) and map that range linearly onto [new_min, new_max]; a constant volume gives zeros"""
    volume = np.asarray(volume)
    intensities = volume[volume > 0] if use_positive_only else volume.reshape(-1)
    robust_min = np.min(intensities) if min_percentile == 0 else np.percentile(intensities, min_percentile)
    robust_max = np.max(intensities) if max_percentile == 100 else np.percentile(intensities, max_percentile)
    if robust_min == robust_max:
        return np.zeros_like(volume)
    return new_min + (np.clip(volume, robust_min, robust_max) - robust_min) / (robust_max - robust_min) * (new_max - new_min)


def get_volume_info(path_volume, return_volume=False, aff_ref=None, max_channels=10):
    im, aff, header = load_volume(path_volume, im_only=False)
    im_shape = list(im.shape)
    n_dims, n_channels = get_dims(im_shape, max_channels=max_channels)
    im_shape = im_shape[:n_dims]
    if '.nii' in path_volume:
        data_res = np.array(header['pixdim'][1:n_dims + 1], dtype=np.float64)
    else:
        data_res = np.array([1.0] * n_dims)
    if aff_ref is not None:
        ras_axes = get_ras_axes(aff, n_dims=n_dims)
        ras_axes_ref = get_ras_axes(aff_ref, n_dims=n_dims)
        im = align_volume_to_ref(im, aff, aff_ref=aff_ref, n_dims=n_dims)
        im_shape = np.array(im_shape)
        data_res = np.array(data_res)
        im_shape[ras_axes_ref] = im_shape[ras_axes]
        data_res[ras_axes_ref] = data_res[ras_axes]
        im_shape = im_shape.tolist()
    if return_volume:
        return im, im_shape, aff, n_dims, n_channels, header, data_res
    return im_shape, aff, n_dims, n_channels, header, data_res


_NEUTRAL_FS = [0, 14, 15, 16, 21, 22, 23, 24, 72, 77, 80, 85, 100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 165, 200,
               201, 202, 203, 204, 205, 206, 207, 208, 209, 210, 251, 252, 253, 254, 255, 258, 259, 260, 331, 332, 333,
               334, 335, 336, 337, 338, 339, 340, 502, 506, 507, 508, 509, 511, 512, 514, 515, 516, 517, 530, 531, 532,
               533, 534, 535, 536, 537]


def get_list_labels(label_list=None, labels_dir=None, save_label_list=None, FS_sort=False):
    """ext/lab2im/utils.py:209-285"""
    if label_list is not None:
        label_list = hm.load_array_if_path(label_list)
        label_list = np.array(hm.reformat_to_list(label_list, dtype='int'))
    elif labels_dir is not None:
        label_list = np.empty(0)
        for path in list_images_in_folder(labels_dir):
            y = load_volume(path, dtype='int32')
            label_list = np.unique(np.concatenate((label_list, np.unique(y)))).astype('int')
    else:
        raise Exception('either label_list, path_label_list or labels_dir should be provided')
    n_neutral_labels = 0
    if FS_sort:
        neutral, left, right = [], [], []
        for la in label_list:
            if la in _NEUTRAL_FS:
                if la not in neutral:
                    neutral.append(la)
            elif (0 < la < 14) | (16 < la < 21) | (24 < la < 40) | (135 < la < 139) | (1000 <= la <= 1035) | \
                    (la == 865) | (20100 < la < 20110):
                if la not in left:
                    left.append(la)
            elif (39 < la < 72) | (162 < la < 165) | (2000 <= la <= 2035) | (20000 < la < 20010) | (la == 139) | \
                    (la == 866):
                if la not in right:
                    right.append(la)
            else:
                raise Exception('label {} not in our current FS classification, '
                                'please update get_list_labels in utils.py'.format(la))
        label_list = np.concatenate([sorted(neutral), sorted(left), sorted(right)])
        if ((len(left) > 0) & (len(right) > 0)) | ((len(left) == 0) & (len(right) == 0)):
            n_neutral_labels = len(neutral)
        else:
            n_neutral_labels = len(label_list)
    if save_label_list is not None:
        np.save(save_label_list, np.int32(label_list))
    if FS_sort:
        return np.int32(label_list), n_neutral_labels
    return np.int32(label_list), None
