#!/usr/bin/env python
"""Command-line launcher of `synthsr_amd.training.training` with the flags of the reference's scripts/training.py:21-93
(same names, destinations and defaults, so existing job scripts keep working).

    python scripts/training.py <labels_dir> <model_dir> <prior_means.npy> <prior_stds.npy> <generation_labels.npy> [flags]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/training.py ...     (N GPUs)"""
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synthsr_amd.training import training  # noqa: E402


def number_bool_or_text(text):
    """values such as --scaling 0.2, --input_channels True, --target_res path.npy (the reference's `infer`,
    ext/lab2im/utils.py:821-832: a float if it parses as one, else the booleans 'True' / 'False', else the string)"""
    try:
        return float(text)
    except ValueError:
        return {'True': True, 'False': False}.get(text, text)


POSITIONAL = ('labels_dir', 'model_dir', 'prior_means', 'prior_stds', 'path_generation_labels')
# (flag, keyword of training() if different, type, default) - the reference's flags, in its order
VALUED = [
    ('images', 'images_dir', str, None), ('generation_classes', 'path_generation_classes', str, None),
    ('prior_distributions', None, str, 'normal'), ('batchsize', None, int, 1), ('input_channels', None, str, True),
    ('output_channel', None, int, None), ('target_res', None, float, None), ('output_shape', None, int, None),
    ('scaling', 'scaling_bounds', number_bool_or_text, 0.15), ('rotation', 'rotation_bounds', number_bool_or_text, 15),
    ('shearing', 'shearing_bounds', number_bool_or_text, .02), ('translation', 'translation_bounds', number_bool_or_text, 5),
    ('nonlin_std', None, float, 4.), ('nonlin_shape_factor', None, float, .03125),
    ('data_res', None, number_bool_or_text, None), ('thickness', None, number_bool_or_text, None),
    ('blur_range', None, float, 1.15), ('bias_std', 'bias_field_std', float, .3), ('bias_shape_factor', None, float, .03125),
    ('n_levels', None, int, 5), ('conv_per_level', 'nb_conv_per_level', int, 2), ('conv_size', None, int, 3),
    ('unet_feat', 'unet_feat_count', int, 24), ('feat_mult', 'feat_multiplier', int, 2), ('dropout', None, float, 0.),
    ('activation', None, str, 'elu'), ('lr', None, float, 1e-4), ('lr_decay', None, float, 0), ('epochs', None, int, 100),
    ('steps_per_epoch', None, int, 1000), ('metric', 'regression_metric', str, 'l1'),
    ('residual_channel', 'work_with_residual_channel', int, None), ('loss_cropping', None, int, None),
    ('checkpoint', None, str, None), ('seg_reg_model_file', 'segmentation_model_file', str, None),
    ('seg_reg_label_list', 'segmentation_label_list', str, None),
    ('seg_reg_leabel_equiv', 'segmentation_label_equivalency', str, None),      # the reference's spelling
    ('seg_reg_rel_weight', 'relative_weight_segmentation', float, 0.25),
]
# switches: (flag, keyword, value stored when the flag is present).  As in the reference, `--downsample` makes the
# command-line default False although training() defaults to True.
SWITCHES = [('no_flipping', 'flipping', False), ('no_reg_error', 'simulate_registration_error', False),
            ('downsample', 'downsample', True), ('no_rel_map', 'build_reliability_maps', False)]
# additional flags of this build (parameters of training() the reference's script does not expose)
EXTRA_VALUED = [('padding_margin', None, int, None), ('seed', None, int, 0)]
EXTRA_SWITCHES = [('no_FS_sort', 'FS_sort', False), ('randomise_res', 'randomise_res', True),
                  ('fs_header_segnet', 'fs_header_segnet', True)]


def build_parser():
    parser = ArgumentParser(description=__doc__.split('\n')[0])
    for name in POSITIONAL:
        parser.add_argument(name, type=str)
    for flag, keyword, kind, default in VALUED + EXTRA_VALUED:
        parser.add_argument('--' + flag, dest=keyword or flag, type=kind, default=default)
    for flag, keyword, stored in SWITCHES + EXTRA_SWITCHES:
        parser.add_argument('--' + flag, dest=keyword, action='store_true' if stored else 'store_false')
    return parser


if __name__ == '__main__':
    training(**vars(build_parser().parse_args()))
