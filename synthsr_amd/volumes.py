"""Volume IO / orientation helpers the callers of the hot path need.  Same names, arguments and results as the reference's
ext/lab2im/utils.py:76-160 (`load_volume`, `save_volume`), :163-206 (`get_volume_info`), :209-285 (`get_list_labels`),
:391-424 (`list_images_in_folder`), ext/lab2im/edit_volumes.py:591-654 (`get_ras_axes`, `align_volume_to_ref`), :504-556
(`resample_volume`) and :148-176 (`rescale_volume`); NIfTI / MGZ through this package's own readers (nifti.py, mgh.py).
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


_VOLUME_EXTENSIONS = ('.nii.gz', '.nii', '.mgz', '.npz')


def list_images_in_folder(path_dir, include_single_image=True, check_if_empty=True):
    """sorted paths of the volumes (.nii, .nii.gz, .mgz, .npz) of a folder; the path of a single volume gives [path]"""
    name = os.path.basename(path_dir)
    if include_single_image and any(ext in name for ext in _VOLUME_EXTENSIONS):
        assert os.path.isfile(path_dir), 'file %s does not exist' % path_dir
        return [path_dir]
    if not os.path.isdir(path_dir):
        raise Exception('Folder does not exist: %s' % path_dir)
    found = sorted(os.path.join(path_dir, f) for f in os.listdir(path_dir)
                   if not f.startswith('.') and f.endswith(('nii.gz', 'nii', '.mgz', '.npz')))
    assert found or not check_if_empty, 'no .nii, .nii.gz, .mgz or .npz image could be found in %s' % path_dir
    return found


def align_volume_to_ref(volume, aff, aff_ref=None, return_aff=False, n_dims=None, return_copy=True):
    """Re-orders and flips the spatial axes of `volume` (affine `aff`) so that they run along the same world axes, in the
    same direction, as those of a volume with affine `aff_ref` (identity = RAS); result of
    ext/lab2im/edit_volumes.py:609-654.  Returns the volume (a view unless return_copy), and its affine if return_aff."""
    out = volume.copy() if return_copy else volume
    aff_new = np.array(aff, dtype=np.float64)
    aff_ref = np.eye(4) if aff_ref is None else np.asarray(aff_ref)
    if n_dims is None:
        n_dims, _ = get_dims(out.shape)
    # get_ras_axes: world axis -> image axis.  Output image axis to[w] must carry what input image axis frm[w] carries.
    to, frm = get_ras_axes(aff_ref, n_dims=n_dims), get_ras_axes(aff_new, n_dims=n_dims)
    order = np.arange(out.ndim)
    order[to] = frm
    out = np.transpose(out, order)
    aff_new[:, to] = aff_new[:, frm].copy()
    # an axis pointing against its reference axis is reversed; the origin moves to the other end of that axis
    for k in np.flatnonzero(np.sum(aff_new[:3, :3] * aff_ref[:3, :3], axis=0)[:n_dims] < 0):
        out = np.flip(out, axis=k)
        aff_new[:, k] = -aff_new[:, k]
        aff_new[:3, 3] -= aff_new[:3, k] * (out.shape[k] - 1)
    return (out, aff_new) if return_aff else out


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
    """([volume,] spatial shape, affine, n_dims, n_channels, header, voxel size) of a volume file; with `aff_ref` the
    shape and voxel size (and the volume) are given in the axis order of that orientation (ext/lab2im/utils.py:163-206)"""
    im, aff, header = load_volume(path_volume, im_only=False)
    n_dims, n_channels = get_dims(list(im.shape), max_channels=max_channels)
    shape = np.array(im.shape[:n_dims])
    if '.nii' in path_volume:
        res = np.array(header['pixdim'][1:n_dims + 1], dtype=np.float64)
    elif '.mgz' in path_volume:                     # ext/lab2im/utils.py:185: header['delta']
        res = np.array(header['delta'][:n_dims], dtype=np.float64)
    else:
        res = np.ones(n_dims)
    if aff_ref is not None:
        here, there = get_ras_axes(aff, n_dims=n_dims), get_ras_axes(aff_ref, n_dims=n_dims)
        im = align_volume_to_ref(im, aff, aff_ref=aff_ref, n_dims=n_dims)
        shape[there], res[there] = shape[here].copy(), res[here].copy()
    info = (shape.tolist(), aff, n_dims, n_channels, header, res)
    return (im,) + info if return_volume else info


_NEUTRAL_FS = [0, 14, 15, 16, 21, 22, 23, 24, 72, 77, 80, 85, 100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 165, 200,
               201, 202, 203, 204, 205, 206, 207, 208, 209, 210, 251, 252, 253, 254, 255, 258, 259, 260, 331, 332, 333,
               334, 335, 336, 337, 338, 339, 340, 502, 506, 507, 508, 509, 511, 512, 514, 515, 516, 517, 530, 531, 532,
               533, 534, 535, 536, 537]


# FreeSurfer label values by side (the classification of ext/lab2im/utils.py:239-262), inclusive ranges
_LEFT_FS = ((1, 13), (17, 20), (25, 39), (136, 138), (1000, 1035), (865, 865), (20101, 20109))
_RIGHT_FS = ((40, 71), (163, 164), (2000, 2035), (20001, 20009), (139, 139), (866, 866))


def _fs_side(label):
    if label in _NEUTRAL_FS:
        return 0
    for side, ranges in ((1, _LEFT_FS), (2, _RIGHT_FS)):
        if any(lo <= label <= hi for lo, hi in ranges):
            return side
    raise Exception('label {} not in our current FS classification, '
                    'please update get_list_labels in utils.py'.format(label))


def get_list_labels(label_list=None, labels_dir=None, save_label_list=None, FS_sort=False):
    """(int32 label values, n_neutral_labels | None): the given list, or every value found in the label maps of
    `labels_dir`; with FS_sort ordered [neutral, left, right] (each ascending), n_neutral counting the whole list when only
    one side is present (ext/lab2im/utils.py:209-285)"""
    if label_list is not None:
        labels = np.array(hm.reformat_to_list(hm.load_array_if_path(label_list), dtype='int'))
    elif labels_dir is not None:
        labels = np.zeros(0, dtype='int')
        for path in list_images_in_folder(labels_dir):
            labels = np.union1d(labels, np.unique(load_volume(path, dtype='int32'))).astype('int')
    else:
        raise Exception('either label_list, path_label_list or labels_dir should be provided')
    n_neutral = None
    if FS_sort:
        sides = np.array([_fs_side(int(v)) for v in labels], dtype=int)
        groups = [np.unique(labels[sides == k]) for k in range(3)]
        labels = np.concatenate(groups)
        one_sided = (len(groups[1]) > 0) != (len(groups[2]) > 0)
        n_neutral = len(labels) if one_sided else len(groups[0])
    if save_label_list is not None:
        np.save(save_label_list, np.int32(labels))
    return np.int32(labels), n_neutral