"""Host input sampler — counterpart of SynthSR/model_inputs.py:25-139 (`build_model_inputs`).

Same generator protocol: yields `[labels [B,*S,1] int32, means [B,L,C], stds [B,L,C]]`.  Differences that
are deliberate: label maps are read from disk ONCE and cached (the reference gunzips a
17 MB NIfTI every step, model_inputs.py:91), and an explicit numpy Generator can be supplied for
reproducible per-rank streams (the reference uses the unseeded global numpy RNG).
"""
import numpy as np

from . import host_math as hm
from . import volumes


def build_model_inputs(path_label_maps,
                       n_labels,
                       prior_means,
                       prior_stds,
                       prior_distributions,
                       path_images=None,
                       batchsize=1,
                       n_channels=1,
                       generation_classes=None,
                       rng=None,
                       label_maps=None):
    """`label_maps` (optional): list of already loaded int32 volumes replacing `path_label_maps`.
    `path_images`: real scans matching the label maps one to one (same order); each batch then carries a 4th input, the
    scan of the picked label map, float [B, *shape, 1] (model_inputs.py:94-96,131-132)."""
    if path_images is not None and len(path_images) != (len(label_maps) if label_maps is not None else len(path_label_maps)):
        raise ValueError('path_images and the label maps should have the same length')
    if generation_classes is None:
        generation_classes = np.arange(n_labels)
    n_classes = len(np.unique(generation_classes))
    npr = np.random if rng is None else rng
    cache = {}
    n_maps = len(label_maps) if label_maps is not None else len(path_label_maps)

    def get_labels(idx):
        if label_maps is not None:
            return label_maps[idx]
        if idx not in cache:
            cache[idx] = volumes.load_volume(path_label_maps[idx], dtype='int', aff_ref=np.eye(4))
        return cache[idx]

    img_cache = {}

    def get_image(idx):
        if idx not in img_cache:
            v = path_images[idx]
            img_cache[idx] = v if isinstance(v, np.ndarray) else volumes.load_volume(v, dtype='float', aff_ref=np.eye(4))
        return img_cache[idx]

    def randint(n, size):
        return npr.randint(n, size=size) if hasattr(npr, 'randint') else npr.integers(n, size=size)

    while True:
        indices = randint(n_maps, batchsize)
        list_label_maps, list_means, list_stds, list_images = [], [], [], []
        for idx in indices:
            lab = get_labels(int(idx))
            list_label_maps.append(lab[np.newaxis, ..., np.newaxis])
            if path_images is not None:
                list_images.append(np.asarray(get_image(int(idx)))[np.newaxis, ..., np.newaxis])
            means = np.empty((1, n_labels, 0))
            stds = np.empty((1, n_labels, 0))
            for channel in range(n_channels):
                if isinstance(prior_means, np.ndarray):
                    if prior_means.shape[0] / 2 != n_channels:
                        raise ValueError("the number of blocks in prior_means does not match n_channels.")
                    tmp_prior_means = prior_means[2 * channel:2 * channel + 2, :]
                else:
                    tmp_prior_means = prior_means
                if isinstance(prior_stds, np.ndarray):
                    if prior_stds.shape[0] / 2 != n_channels:
                        raise ValueError("the number of blocks in prior_stds does not match n_channels.")
                    tmp_prior_stds = prior_stds[2 * channel:2 * channel + 2, :]
                else:
                    tmp_prior_stds = prior_stds
                tmp_classes_means = hm.draw_value_from_distribution(tmp_prior_means, n_classes, prior_distributions,
                                                                    125., 100., positive_only=True, rng=rng)
                tmp_classes_stds = hm.draw_value_from_distribution(tmp_prior_stds, n_classes, prior_distributions,
                                                                   15., 10., positive_only=True, rng=rng)
                tmp_means = tmp_classes_means[generation_classes][np.newaxis, :, np.newaxis]
                tmp_stds = tmp_classes_stds[generation_classes][np.newaxis, :, np.newaxis]
                means = np.concatenate([means, tmp_means], axis=-1)
                stds = np.concatenate([stds, tmp_stds], axis=-1)
            list_means.append(means)
            list_stds.append(stds)
        list_inputs = [list_label_maps, list_means, list_stds]
        if path_images is not None:
            list_inputs.append(list_images)
        if batchsize > 1:
            list_inputs = [np.concatenate(item, 0) for item in list_inputs]
        else:
            list_inputs = [item[0] for item in list_inputs]
        yield list_inputs
