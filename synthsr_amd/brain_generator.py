"""`BrainGenerator` — same constructor signature, attributes and `generate_brain()` as
SynthSR/brain_generator.py:28-330, backed by the HIP generator (labels_to_image_model.py)."""
import numpy as np

from . import host_math as hm
from . import volumes
from .labels_to_image_model import labels_to_image_model
from .model_inputs import build_model_inputs


class BrainGenerator:

    def __init__(self,
                 labels_dir,
                 prior_means,
                 prior_stds,
                 prior_distributions,
                 generation_labels,
                 images_dir=None,
                 n_neutral_labels=None,
                 padding_margin=None,
                 batchsize=1,
                 input_channels=1,
                 output_channel=0,
                 target_res=None,
                 output_shape=None,
                 output_div_by_n=None,
                 generation_classes=None,
                 flipping=True,
                 scaling_bounds=0.15,
                 rotation_bounds=15,
                 shearing_bounds=.012,
                 translation_bounds=5,
                 nonlin_std=3.,
                 nonlin_shape_factor=0.0625,
                 simulate_registration_error=True,
                 randomise_res=False,
                 data_res=None,
                 thickness=None,
                 downsample=False,
                 blur_range=1.15,
                 build_reliability_maps=False,
                 bias_field_std=0.3,
                 bias_shape_factor=0.025,
                 device=None,
                 rng=None,
                 label_maps=None):
        """Parameters as in the reference (brain_generator.py:62-191).  Extra, optional:
        device (torch device), rng (numpy Generator for the host input sampler), label_maps (list of int32
        volumes already in memory, used instead of reading `labels_dir`; needs labels_dir=None)."""
        if label_maps is not None:
            self.labels_paths = None
            self.label_maps = [np.ascontiguousarray(m, dtype=np.int32) for m in label_maps]
            self.labels_shape = list(self.label_maps[0].shape)
            self.aff, self.n_dims, self.header, self.atlas_res = np.eye(4), 3, None, np.array([1., 1., 1.])
        else:
            self.labels_paths = volumes.list_images_in_folder(labels_dir)
            self.label_maps = None
            self.labels_shape, self.aff, self.n_dims, _, self.header, self.atlas_res = \
                volumes.get_volume_info(self.labels_paths[0], aff_ref=np.eye(4))
        # real scans as regression targets (brain_generator.py:197-198): a folder, or a list of in-memory float volumes
        if images_dir is None:
            self.images_paths = None
        elif isinstance(images_dir, (list, tuple)):
            self.images_paths = [np.asarray(v, dtype=np.float32) for v in images_dir]
        else:
            self.images_paths = volumes.list_images_in_folder(images_dir)
        if generation_labels is not None:
            self.generation_labels = np.asarray(hm.load_array_if_path(generation_labels))
        else:
            self.generation_labels, _ = volumes.get_list_labels(labels_dir=labels_dir)
        self.n_neutral_labels = n_neutral_labels if n_neutral_labels is not None else self.generation_labels.shape[0]
        self.input_channels = np.array(hm.reformat_to_list(input_channels))
        self.output_channel = hm.reformat_to_list(output_channel)
        self.n_channels = len(self.input_channels)
        self.target_res = hm.load_array_if_path(target_res)
        self.batchsize = batchsize
        self.padding_margin = hm.load_array_if_path(padding_margin)
        self.flipping = flipping
        self.output_shape = hm.load_array_if_path(output_shape)
        self.output_div_by_n = output_div_by_n
        self.prior_distributions = prior_distributions
        if generation_classes is not None:
            self.generation_classes = np.asarray(hm.load_array_if_path(generation_classes))
            assert self.generation_classes.shape == self.generation_labels.shape, \
                'if provided, generation_classes should have the same shape as generation_labels'
            unique_classes = np.unique(self.generation_classes)
            assert np.array_equal(unique_classes, np.arange(np.max(unique_classes) + 1)), \
                'generation_classes should a linear range between 0 and its maximum value.'
        else:
            self.generation_classes = np.arange(self.generation_labels.shape[0])
        self.prior_means = hm.load_array_if_path(prior_means)
        self.prior_stds = hm.load_array_if_path(prior_stds)
        self.scaling_bounds = hm.load_array_if_path(scaling_bounds)
        self.rotation_bounds = hm.load_array_if_path(rotation_bounds)
        self.shearing_bounds = hm.load_array_if_path(shearing_bounds)
        self.translation_bounds = hm.load_array_if_path(translation_bounds)
        self.nonlin_std = nonlin_std
        self.nonlin_shape_factor = nonlin_shape_factor
        self.simulate_registration_error = simulate_registration_error
        # the reference evaluates `None & bool` here when training() passes its default randomise_res=None
        # (F8, TypeError); the intended meaning is False
        self.randomise_res = bool(randomise_res) if randomise_res is not None else False
        self.data_res = hm.load_array_if_path(data_res)
        assert not (self.randomise_res & (self.data_res is not None)), \
            'randomise_res and data_res cannot be provided at the same time'
        self.thickness = hm.load_array_if_path(thickness)
        self.downsample = downsample
        self.blur_range = blur_range
        self.build_reliability_maps = build_reliability_maps
        self.bias_field_std = bias_field_std
        self.bias_shape_factor = bias_shape_factor
        self.device = device
        self.rng = rng

        self.labels_to_image_model, self.model_output_shape = self._build_labels_to_image_model()
        self.model_inputs_generator = self._build_model_inputs_generator()
        self.brain_generator = self._build_brain_generator()

    def _build_labels_to_image_model(self):
        m = labels_to_image_model(labels_shape=self.labels_shape,
                                  input_channels=self.input_channels,
                                  output_channel=self.output_channel,
                                  generation_labels=self.generation_labels,
                                  n_neutral_labels=self.n_neutral_labels,
                                  atlas_res=self.atlas_res,
                                  target_res=self.target_res,
                                  output_shape=self.output_shape,
                                  output_div_by_n=self.output_div_by_n,
                                  padding_margin=self.padding_margin,
                                  flipping=self.flipping,
                                  aff=np.eye(4),
                                  scaling_bounds=self.scaling_bounds,
                                  rotation_bounds=self.rotation_bounds,
                                  shearing_bounds=self.shearing_bounds,
                                  translation_bounds=self.translation_bounds,
                                  nonlin_std=self.nonlin_std,
                                  nonlin_shape_factor=self.nonlin_shape_factor,
                                  simulate_registration_error=self.simulate_registration_error,
                                  randomise_res=self.randomise_res,
                                  data_res=self.data_res,
                                  thickness=self.thickness,
                                  downsample=self.downsample,
                                  build_reliability_maps=self.build_reliability_maps,
                                  blur_range=self.blur_range,
                                  bias_field_std=self.bias_field_std,
                                  bias_shape_factor=self.bias_shape_factor,
                                  device=self.device)
        return m, m.model_output_shape

    def _build_model_inputs_generator(self):
        return build_model_inputs(path_label_maps=self.labels_paths,
                                  n_labels=len(self.generation_labels),
                                  prior_means=self.prior_means,
                                  prior_stds=self.prior_stds,
                                  prior_distributions=self.prior_distributions,
                                  path_images=self.images_paths,
                                  batchsize=self.batchsize,
                                  n_channels=self.n_channels,
                                  generation_classes=self.generation_classes,
                                  rng=self.rng,
                                  label_maps=self.label_maps)

    def _build_brain_generator(self):
        while True:
            model_inputs = next(self.model_inputs_generator)
            [image, target] = self.labels_to_image_model.predict(model_inputs)
            yield image, target

    def generate_brain(self):
        """-> (image, target) numpy arrays in the native orientation of the label maps, squeezed"""
        (image, target) = next(self.brain_generator)
        list_images, list_targets = [], []
        for i in range(self.batchsize):
            list_images.append(volumes.align_volume_to_ref(image[i], np.eye(4), aff_ref=self.aff, n_dims=self.n_dims))
            list_targets.append(volumes.align_volume_to_ref(target[i], np.eye(4), aff_ref=self.aff,
                                                            n_dims=self.n_dims))
        image = np.squeeze(np.stack(list_images, axis=0))
        target = np.squeeze(np.stack(list_targets, axis=0))
        return image, target
