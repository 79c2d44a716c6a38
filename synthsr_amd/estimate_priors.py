"""Offline estimation of the GMM prior hyper-parameters (`prior_means.npy` / `prior_stds.npy`, the files `training()`
takes) from real scans and their segmentations - SynthSR/estimate_priors.py:76-310, same function names, arguments,
return layouts and errors.  Host-side numpy like the reference's (it is an offline tool, not part of
the per-step path); instead of one boolean scan of the volume per label, voxels are bucketed by class once (LUT +
stable sort), so the cost is one sort per image instead of `len(labels_list)` passes.

`estimate_t2_cropping` (estimate_priors.py:27-73, hippocampus-specific cropping statistics) is not part of this build."""
import os

import numpy as np

from . import volumes as V
from .host_math import reformat_to_list, load_array_if_path

MAD_SCALE = 1.4826   # scipy 1.4.1 (reference requirements.txt:53) median_absolute_deviation default: scale * MAD


def _labels_and_classes(labels_list, classes_list, what='labels'):
    labels_list = np.array(reformat_to_list(load_array_if_path(labels_list), dtype='int'))
    if classes_list is not None:
        classes_list = np.array(reformat_to_list(load_array_if_path(classes_list), dtype='int'))
    else:
        classes_list = np.arange(labels_list.shape[0])
    assert len(classes_list) == len(labels_list), '%s and classes lists should have the same length' % what
    unique_classes = np.unique(classes_list)
    n_classes = len(unique_classes)
    if not np.array_equal(unique_classes, np.arange(n_classes)):
        raise ValueError('classes_list should only contain values between 0 and K-1, '
                         'where K is the total number of classes. Here K = %d' % n_classes)
    return labels_list, classes_list, n_classes


def _median_and_mad(x):
    """np.nanmedian(x) and scipy-1.4.1 median_absolute_deviation(x, nan_policy='omit') of a non-empty 1-D float64 array
    (omit: when the sum is NaN every non-finite value is dropped, as numpy.ma.masked_invalid does)"""
    med = np.nanmedian(x)
    if np.isnan(np.sum(x)):
        x = x[np.isfinite(x)]
        if x.size == 0:
            return med, np.nan
    return med, MAD_SCALE * np.median(np.abs(x - np.median(x)))


def sample_intensity_stats_from_image(image, segmentation, labels_list, classes_list=None, keep_strictly_positive=True):
    """(2, K): row 0 the median intensity of each class, row 1 its scaled median absolute deviation; classes without
    voxels stay 0.  For every class but 0 (background) only strictly positive intensities count when
    keep_strictly_positive (estimate_priors.py:76-130)."""
    labels_list, classes_list, n_classes = _labels_and_classes(labels_list, classes_list)
    image = np.asarray(image)
    seg = np.asarray(segmentation)
    if image.shape != seg.shape:
        raise ValueError('image and segmentation should have the same shape, had %s and %s' % (image.shape, seg.shape))
    seg = np.round(seg).astype(np.int64).reshape(-1)
    lo, hi = int(min(seg.min(), labels_list.min())), int(max(seg.max(), labels_list.max()))
    lut = np.empty(hi - lo + 1, dtype=np.int64)
    flat = image.reshape(-1)
    cls_parts, val_parts = [], []
    remaining = np.arange(len(labels_list))
    while remaining.size:          # one pass; more only if a label is listed several times (it then counts once per
        _, first = np.unique(labels_list[remaining], return_index=True)   # listing, like the reference's per-label scans)
        idx = remaining[first]
        lut.fill(-1)
        lut[labels_list[idx] - lo] = classes_list[idx]
        c = lut[seg - lo]
        keep = c >= 0
        cls_parts.append(c[keep])
        val_parts.append(flat[keep].astype(np.float64))
        remaining = np.delete(remaining, first)
    cls, vals = np.concatenate(cls_parts), np.concatenate(val_parts)
    order = np.argsort(cls, kind='stable')
    cls, vals = cls[order], vals[order]
    starts = np.searchsorted(cls, np.arange(n_classes + 1))
    means, stds = np.zeros(n_classes), np.zeros(n_classes)
    for k in range(n_classes):
        x = vals[starts[k]:starts[k + 1]]
        if k and keep_strictly_positive:
            x = x[x > 0]
        if x.size:
            means[k], stds[k] = _median_and_mad(x)
    return np.stack([means, stds])


def sample_intensity_stats_from_single_dataset(image_dir, labels_dir, labels_list, classes_list=None, max_channel=3,
                                               rescale=True):
    """prior_means, prior_stds of one dataset, each (2*n_channels, K): per channel a row with the mean and a row with
    the std over the images of the per-image class median (prior_means) / class MAD (prior_stds)
    (estimate_priors.py:133-221).  Images and label maps are matched by sorted file order."""
    path_images = V.list_images_in_folder(image_dir)
    path_labels = V.list_images_in_folder(labels_dir)
    assert len(path_images) == len(path_labels), 'image and labels folders do not have the same number of files'
    labels_list, classes_list, n_classes = _labels_and_classes(labels_list, classes_list)
    _, n_channels = V.get_dims(V.load_volume(path_images[0]).shape, max_channels=max_channel)
    means = np.zeros((len(path_images), n_classes, n_channels))
    stds = np.zeros((len(path_images), n_classes, n_channels))
    for idx, (path_im, path_la) in enumerate(zip(path_images, path_labels)):
        image = V.load_volume(path_im)
        la = V.load_volume(path_la)
        if n_channels == 1:
            image = image[..., None]
        for channel in range(n_channels):
            im = image[..., channel]
            if rescale:
                im = V.rescale_volume(im)
            stats = sample_intensity_stats_from_image(im, la, labels_list, classes_list=classes_list)
            means[idx, :, channel] = stats[0]
            stds[idx, :, channel] = stats[1]
    prior_means = np.zeros((2 * n_channels, n_classes))
    prior_stds = np.zeros((2 * n_channels, n_classes))
    prior_means[0::2] = np.mean(means, axis=0).T
    prior_means[1::2] = np.std(means, axis=0).T
    prior_stds[0::2] = np.mean(stds, axis=0).T
    prior_stds[1::2] = np.std(stds, axis=0).T
    return prior_means, prior_stds


def build_intensity_stats(list_image_dir, list_labels_dir, result_dir, estimation_labels, estimation_classes=None,
                          max_channel=3, rescale=True):
    """Several datasets (folders; every channel of a multi-modal folder is its own block of two rows) stacked into
    `result_dir/prior_means.npy` and `prior_stds.npy`, shape (2*n_blocks, K) (estimate_priors.py:224-310).  One labels
    folder may serve all image folders."""
    os.makedirs(result_dir, exist_ok=True)
    list_image_dir = reformat_to_list(list_image_dir)
    list_labels_dir = reformat_to_list(list_labels_dir, length=len(list_image_dir))
    estimation_labels, estimation_classes, _ = _labels_and_classes(estimation_labels, estimation_classes,
                                                                   'estimation labels')
    blocks = [sample_intensity_stats_from_single_dataset(image_dir, labels_dir, estimation_labels, estimation_classes,
                                                         max_channel=max_channel, rescale=rescale)
              for image_dir, labels_dir in zip(list_image_dir, list_labels_dir)]
    prior_means = np.concatenate([b[0] for b in blocks], axis=0)
    prior_stds = np.concatenate([b[1] for b in blocks], axis=0)
    np.save(os.path.join(result_dir, 'prior_means.npy'), prior_means)
    np.save(os.path.join(result_dir, 'prior_stds.npy'), prior_stds)
    return prior_means, prior_stds
