#!/usr/bin/env python
"""Super-resolution / synthesis of 1 mm MP-RAGE volumes on the MI355X: the command line of the reference's
scripts/predict_command_line.py (same positional arguments and flags).

    python scripts/predict_command_line.py <path_images> <path_predictions> [--ct] [--model M.h5|M.npz] [--disable_flipping]

<path_images> and <path_predictions> are both single files or both folders.  `--cpu` and `--threads` exist for
command-line compatibility only: this build runs the U-Net through its HIP kernels."""
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_parser():
    p = ArgumentParser(description='SynthSR prediction (MI355X build)')
    p.add_argument('path_images', help='scan, or folder of scans, to super-resolve')
    p.add_argument('path_predictions', help='output file (or folder, if path_images is a folder)')
    p.add_argument('--ct', action='store_true', help='the inputs are CT scans (clipped to [0, 80] HU first)')
    p.add_argument('--model', default=None, help='weights: Keras .h5 as released with the reference, or .npz checkpoint')
    p.add_argument('--disable_flipping', action='store_true', help='no left-right flip averaging at test time')
    p.add_argument('--cpu', action='store_true', help='reference flag; CPU inference is not part of this build')
    p.add_argument('--threads', type=int, default=1, help='reference flag; ignored')
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.cpu:
        raise NotImplementedError('--cpu: the MI355X build runs the U-Net through its HIP kernels only')
    from synthsr_amd.predict import predict
    print('SynthSR prediction' + ('' if args.model is None else ' with the model ' + args.model))
    predict(args.path_images, args.path_predictions, path_model=args.model, ct=args.ct,
            disable_flipping=args.disable_flipping)


if __name__ == '__main__':
    main()
