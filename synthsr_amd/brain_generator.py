"""`BrainGenerator`: the user-facing wrapper of the synthetic-scan generator.  Public surface (constructor parameters and
their defaults, attributes `labels_to_image_model`, `model_output_shape`, `model_inputs_generator`, `aff`, `header`,
`n_dims`, `generation_labels`, ..., method `generate_brain()`) follows SynthSR/brain_generator.py:28-330 so that scripts
written against the reference keep working; the work is done by the HIP generator in labels_to_image_model.py."""
import numpy as np

from . import host_math as hm
from . import volumes
from .labels_to_image_model import labels_to_image_model
from .model_inputs import build_model_inputs

# constructor arguments kept as attributes unchanged / after `load_array_if_path` (a path to a .npy is accepted, :62-191)
_PLAIN = ('batchsize', 'flipping', 'output_div_by_n', 'prior_distributions', 'nonlin_std', 'nonlin_shape_factor',
          'simulate_registration_error', 'downsample', 'blur_range', 'build_reliability_maps', 'bias_field_std',
          'bias_shape_factor', 'device', 'rng')
_ARRAY_OR_PATH = ('target_res', 'padding_margin', 'output_shape', 'prior_means', 'prior_stds', 'scaling_bounds',
                  'rotation_bounds', 'shearing_bounds', 'translation_bounds', 'data_res', 'thickness')
# what labels_to_image_model() receives straight from the attributes of the same name
_MODEL_ARGS = ('labels_shape', 'input_channels', 'output_channel', 'generation_labels', 'n_neutral_labels', 'atlas_res',
               'target_res', 'output_shape', 'output_div_by_n', 'padding_margin', 'flipping', 'scaling_bounds',
               'rotation_bounds', 'shearing_bounds', 'translation_bounds', 'nonlin_std', 'nonlin_shape_factor',
               'simulate_registration_error', 'randomise_res', 'data_res', 'thickness', 'downsample',
               'build_reliability_maps', 'blur_range', 'bias_field_std', 'bias_shape_factor', 'device')


class BrainGenerator:
    def __init__(self, labels_dir, prior_means, prior_stds, prior_distributions, generation_labels, images_dir=None,
                 n_neutral_labels=None, padding_margin=None, batchsize=1, input_channels=1, output_channel=0,
                 target_res=None, output_shape=None, output_div_by_n=None, generation_classes=None, flipping=True,
                 scaling_bounds=0.15, rotation_bounds=15, shearing_bounds=.012, translation_bounds=5, nonlin_std=3.,
                 nonlin_shape_factor=0.0625, simulate_registration_error=True, randomise_res=False, data_res=None,
                 thickness=None, downsample=False, blur_range=1.15, build_reliability_maps=False, bias_field_std=0.3,
                 bias_shape_factor=0.025, device=None, rng=None, label_maps=None):
        """Parameters: see the reference docstring (brain_generator.py:62-191).  Additional, optional: `device` (torch
        device), `rng` (numpy Generator of the host input sampler), `label_maps` (int32 volumes already in memory, used
        instead of reading `labels_dir`, which must then be None); `images_dir` may also be a list of in-memory scans."""
        given = locals()
        for name in _PLAIN:
            setattr(self, name, given[name])
        for name in _ARRAY_OR_PATH:
            setattr(self, name, hm.load_array_if_path(given[name]))

        # geometry of the label maps; everything downstream works in the RAS frame (aff_ref = identity)
        if label_maps is None:
            self.label_maps, self.labels_paths = None, volumes.list_images_in_folder(labels_dir)
            info = volumes.get_volume_info(self.labels_paths[0], aff_ref=np.eye(4))
            self.labels_shape, self.aff, self.n_dims, self.header, self.atlas_res = info[0], info[1], info[2], info[4], info[5]
        else:
            self.labels_paths = None
            self.label_maps = [np.ascontiguousarray(m, dtype=np.int32) for m in label_maps]
            self.labels_shape, self.n_dims, self.header = list(self.label_maps[0].shape), 3, None
            self.aff, self.atlas_res = np.eye(4), np.ones(3)
        # real scans as regression targets (:197-198)
        if isinstance(images_dir, (list, tuple)):
            self.images_paths = [np.asarray(v, dtype=np.float32) for v in images_dir]
        else:
            self.images_paths = None if images_dir is None else volumes.list_images_in_folder(images_dir)

        # labels, classes, channels
        self.generation_labels = (volumes.get_list_labels(labels_dir=labels_dir)[0] if generation_labels is None
                                  else np.asarray(hm.load_array_if_path(generation_labels)))
        n_labels = self.generation_labels.shape[0]
        self.n_neutral_labels = n_labels if n_neutral_labels is None else n_neutral_labels
        if generation_classes is None:
            self.generation_classes = np.arange(n_labels)
        else:
            self.generation_classes = np.asarray(hm.load_array_if_path(generation_classes))
            assert self.generation_classes.shape == self.generation_labels.shape, \
                'if provided, generation_classes should have the same shape as generation_labels'
            present = np.unique(self.generation_classes)
            assert np.array_equal(present, np.arange(present.max() + 1)), \
                'generation_classes should a linear range between 0 and its maximum value.'
        self.input_channels = np.array(hm.reformat_to_list(input_channels))
        self.output_channel = hm.reformat_to_list(output_channel)
        self.n_channels = len(self.input_channels)
        # training() hands its default randomise_res=None down here, where the reference evaluates `None & bool`
        # (TypeError); the intended meaning is False
        self.randomise_res = bool(randomise_res)
        assert not (self.randomise_res and self.data_res is not None), \
            'randomise_res and data_res cannot be provided at the same time'

        model = labels_to_image_model(aff=np.eye(4), **{name: getattr(self, name) for name in _MODEL_ARGS})
        self.labels_to_image_model, self.model_output_shape = model, model.model_output_shape
        self.model_inputs_generator = build_model_inputs(
            path_label_maps=self.labels_paths, n_labels=n_labels, prior_means=self.prior_means, prior_stds=self.prior_stds,
            prior_distributions=self.prior_distributions, path_images=self.images_paths, batchsize=self.batchsize,
            n_channels=self.n_channels, generation_classes=self.generation_classes, rng=self.rng,
            label_maps=self.label_maps)
        self.brain_generator = self._batches()

    def _batches(self):
        """endless stream of (image, target) batches in the RAS frame"""
        for model_inputs in self.model_inputs_generator:
            image, target = self.labels_to_image_model.predict(model_inputs)
            yield image, target

    def generate_brain(self):
        """one batch -> (image, target) numpy arrays, each item re-oriented from RAS to the native orientation of the
        label maps (brain_generator.py:317-330), singleton axes squeezed"""
        image, target = next(self.brain_generator)

        def native(batch):
            items = [volumes.align_volume_to_ref(batch[i], np.eye(4), aff_ref=self.aff, n_dims=self.n_dims)
                     for i in range(self.batchsize)]
            return np.squeeze(np.stack(items, axis=0))
        return native(image), native(target)
