"""Host input sampler: picks label maps and draws the per-class GMM parameters of every batch - the job of
SynthSR/model_inputs.py:25-139 (`build_model_inputs`), same generator protocol: each `next()` gives
`[labels [B,*S,1] int32, means [B,L,C], stds [B,L,C]]` (+ `image [B,*S,1]` when real scans are given).

Deliberate differences: label maps and scans are read from disk once and kept (the reference gunzips a
17 MB NIfTI every step, model_inputs.py:91), in-memory volumes are accepted, and an explicit numpy Generator gives
reproducible per-rank streams (the reference draws from the unseeded global numpy state)."""
import numpy as np

from . import host_math as hm
from . import volumes

# defaults of the class means / standard deviations when no prior is given (model_inputs.py:118-121)
_MEAN_CENTRE, _MEAN_RANGE, _STD_CENTRE, _STD_RANGE = 125., 100., 15., 10.


class _Lazy:
    """volumes addressed by index: arrays are used as they are, paths are loaded (RAS frame) on first use and kept"""

    def __init__(self, items, dtype):
        self.items, self.dtype, self.loaded = items, dtype, {}

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        item = self.items[i]
        if isinstance(item, np.ndarray):
            return item
        if i not in self.loaded:
            self.loaded[i] = volumes.load_volume(item, dtype=self.dtype, aff_ref=np.eye(4))
        return self.loaded[i]


def _channel_prior(prior, channel, n_channels, name):
    """rows [2c, 2c+2) of a stacked (2*n_channels, K) prior array; anything else (None, number, pair) serves all channels"""
    if not isinstance(prior, np.ndarray):
        return prior
    if prior.shape[0] / 2 != n_channels:
        raise ValueError("the number of blocks in %s does not match n_channels." % name)
    return prior[2 * channel:2 * channel + 2, :]


def build_model_inputs(path_label_maps, n_labels, prior_means, prior_stds, prior_distributions, path_images=None,
                       batchsize=1, n_channels=1, generation_classes=None, rng=None, label_maps=None):
    """`label_maps` (optional): already loaded int32 volumes replacing `path_label_maps`.  `path_images`: real scans (paths
    or arrays) matching the label maps one to one, in the same order (model_inputs.py:94-96,131-132)."""
    maps = _Lazy(label_maps if label_maps is not None else path_label_maps, 'int')
    scans = None if path_images is None else _Lazy(path_images, 'float')
    if scans is not None and len(scans) != len(maps):
        raise ValueError('path_images and the label maps should have the same length')
    classes = np.arange(n_labels) if generation_classes is None else generation_classes
    n_classes = len(np.unique(classes))
    legacy = rng is None or hasattr(rng, 'randint')           # numpy's global state / RandomState vs Generator
    source = np.random if rng is None else rng

    def class_stats():
        """[1, L, C] means and stds: one draw per class and channel, spread over the labels of the class"""
        cols_m, cols_s = [], []
        for c in range(n_channels):
            m = hm.draw_value_from_distribution(_channel_prior(prior_means, c, n_channels, 'prior_means'), n_classes,
                                                prior_distributions, _MEAN_CENTRE, _MEAN_RANGE, positive_only=True, rng=rng)
            s = hm.draw_value_from_distribution(_channel_prior(prior_stds, c, n_channels, 'prior_stds'), n_classes,
                                                prior_distributions, _STD_CENTRE, _STD_RANGE, positive_only=True, rng=rng)
            cols_m.append(m[classes])
            cols_s.append(s[classes])
        if not cols_m:
            return np.empty((1, n_labels, 0)), np.empty((1, n_labels, 0))
        return np.stack(cols_m, -1)[None], np.stack(cols_s, -1)[None]

    def batches(state):
        while True:
            picks = source.randint(len(maps), size=batchsize) if legacy else source.integers(len(maps), size=batchsize)
            batch = [[], [], []] + ([[]] if scans is not None else [])
            for i in (int(p) for p in picks):
                batch[0].append(maps[i][None, ..., None])
                if scans is not None:
                    batch[3].append(np.asarray(scans[i])[None, ..., None])
                means, stds = class_stats()
                batch[1].append(means)
                batch[2].append(stds)
            state.last_picks = [int(p) for p in picks]
            yield [np.concatenate(items, 0) if batchsize > 1 else items[0] for items in batch]

    return ModelInputs(batches, len(maps), scans is not None)


class ModelInputs:
    """the generator object build_model_inputs returns: the reference's protocol (`next()` -> [labels, means, stds(, image)],
    model_inputs.py:86-139) plus `last_picks`, the indices of the label maps of the batch just produced -- what lets
    training() keep the maps it has already used on the device and pick from that pool instead of copying 16 MB per step"""

    def __init__(self, batches, n_maps, has_images):
        self.last_picks, self.n_maps, self.has_images = None, n_maps, has_images
        self._it = batches(self)

    def __iter__(self):
        return self

    def __next__(self):
        return next(self._it)
