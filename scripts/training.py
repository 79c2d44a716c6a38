#!/usr/bin/env python
"""Command-line launcher with the flags of the reference's scripts/training.py:21-93.
Multi-GPU: python -m torch.distributed.run --nproc-per-node N scripts/training.py ..."""
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synthsr_amd.training import training  # noqa: E402


def infer(x):
    """ext/lab2im/utils.py:821-832: float, then bool, else str"""
    try:
        return float(x)
    except ValueError:
        if x == 'False':
            return False
        if x == 'True':
            return True
        if not isinstance(x, str):
            raise TypeError('input should be an int/float/boolean/str, had {}'.format(type(x)))
        return x


parser = ArgumentParser()
parser.add_argument("labels_dir", type=str)
parser.add_argument("model_dir", type=str)
parser.add_argument("prior_means", type=str)
parser.add_argument("prior_stds", type=str)
parser.add_argument("path_generation_labels", type=str)
parser.add_argument("--prior_distributions", type=str, dest="prior_distributions", default='normal')
parser.add_argument("--images_dir", type=str, dest="images_dir", default=None)
parser.add_argument("--generation_classes", type=str, dest="path_generation_classes", default=None)
parser.add_argument("--no_FS_sort", action='store_false', dest="FS_sort")
parser.add_argument("--batchsize", type=int, dest="batchsize", default=1)
parser.add_argument("--input_channels", type=infer, dest="input_channels", default=True)
parser.add_argument("--output_channel", type=int, dest="output_channel", default=0)
parser.add_argument("--target_res", type=infer, dest="target_res", default=None)
parser.add_argument("--output_shape", type=int, dest="output_shape", default=None)
parser.add_argument("--no_flipping", action='store_false', dest="flipping")
parser.add_argument("--padding_margin", type=int, dest="padding_margin", default=None)
parser.add_argument("--scaling", type=infer, dest="scaling_bounds", default=0.15)
parser.add_argument("--rotation", type=infer, dest="rotation_bounds", default=15)
parser.add_argument("--shearing", type=infer, dest="shearing_bounds", default=.02)
parser.add_argument("--translation", type=infer, dest="translation_bounds", default=5)
parser.add_argument("--nonlin_std", type=float, dest="nonlin_std", default=4.)
parser.add_argument("--nonlin_shape_factor", type=float, dest="nonlin_shape_factor", default=.03125)
parser.add_argument("--no_simulate_registration_error", action='store_false', dest="simulate_registration_error")
parser.add_argument("--data_res", type=infer, dest="data_res", default=None)
parser.add_argument("--thickness", type=infer, dest="thickness", default=None)
parser.add_argument("--randomise_res", action='store_true', dest="randomise_res")
parser.add_argument("--no_downsample", action='store_false', dest="downsample")
parser.add_argument("--blur_range", type=float, dest="blur_range", default=1.15)
parser.add_argument("--no_reliability_maps", action='store_false', dest="build_reliability_maps")
parser.add_argument("--bias_std", type=float, dest="bias_field_std", default=.3)
parser.add_argument("--bias_shape_factor", type=float, dest="bias_shape_factor", default=.03125)
parser.add_argument("--n_levels", type=int, dest="n_levels", default=5)
parser.add_argument("--conv_per_level", type=int, dest="nb_conv_per_level", default=2)
parser.add_argument("--conv_size", type=int, dest="conv_size", default=3)
parser.add_argument("--unet_feat", type=int, dest="unet_feat_count", default=24)
parser.add_argument("--feat_mult", type=int, dest="feat_multiplier", default=2)
parser.add_argument("--dropout", type=float, dest="dropout", default=0.)
parser.add_argument("--activation", type=str, dest="activation", default='elu')
parser.add_argument("--lr", type=float, dest="lr", default=1e-4)
parser.add_argument("--lr_decay", type=float, dest="lr_decay", default=0)
parser.add_argument("--epochs", type=int, dest="epochs", default=100)
parser.add_argument("--steps_per_epoch", type=int, dest="steps_per_epoch", default=1000)
parser.add_argument("--regression_metric", type=str, dest="regression_metric", default='l1')
parser.add_argument("--work_with_residual_channel", type=int, dest="work_with_residual_channel", default=None)
parser.add_argument("--loss_cropping", type=int, dest="loss_cropping", default=None)
parser.add_argument("--checkpoint", type=str, dest="checkpoint", default=None)
parser.add_argument("--seed", type=int, dest="seed", default=0)

if __name__ == '__main__':
    training(**vars(parser.parse_args()))
