#!/usr/bin/env python
"""Low-field (Hyperfine) variant: a T1 and a T2 acquisition at 1.5 x 1.5 x 5 mm in, one synthetic 1 mm MP-RAGE out - the
command line of the reference's scripts/predict_command_line_hyperfine.py.

    python scripts/predict_command_line_hyperfine.py <path_t1_images> <path_t2_images> <path_predictions> [--model M]

The three paths are all single files or all folders (T1 and T2 matched by sorted file name).  `--cpu` / `--threads` exist
for command-line compatibility only."""
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_parser():
    p = ArgumentParser(description='SynthSR-hyperfine prediction (MI355X build)')
    for name, what in (('path_t1_images', 'T1 scan(s) at the native axial resolution'),
                       ('path_t2_images', 'matching T2 scan(s)'), ('path_predictions', 'output file or folder')):
        p.add_argument(name, help=what)
    p.add_argument('--model', default=None, help='weights: Keras .h5 as released with the reference, or .npz checkpoint')
    p.add_argument('--cpu', action='store_true', help='reference flag; CPU inference is not part of this build')
    p.add_argument('--threads', type=int, default=1, help='reference flag; ignored')
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.cpu:
        raise NotImplementedError('--cpu: the MI355X build runs the U-Net through its HIP kernels only')
    from synthsr_amd.predict import predict_hyperfine
    print('SynthSR-hyperfine prediction')
    predict_hyperfine(args.path_t1_images, args.path_t2_images, args.path_predictions, path_model=args.model)
    print('All done!')


if __name__ == '__main__':
    main()
