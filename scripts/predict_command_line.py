#!/usr/bin/env python
"""MI355X build of the reference's scripts/predict_command_line.py (same positional arguments and flags):
    python scripts/predict_command_line.py <path_images> <path_predictions> [--ct] [--model M] [--disable_flipping]
--cpu / --threads are accepted for command-line compatibility; this build has no CPU inference path."""
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

parser = ArgumentParser()
parser.add_argument("path_images", type=str,
                    help="images to super-resolve / synthesize. Can be the path to a single image or to a folder")
parser.add_argument("path_predictions", type=str,
                    help="path where to save the synthetic 1mm MP-RAGEs. Must be the same type "
                         "as path_images (path to a single image or to a folder)")
parser.add_argument("--cpu", action="store_true", help="(reference flag) CPU inference: not available in this build.")
parser.add_argument("--threads", type=int, default=1, dest="threads", help="(reference flag) ignored.")
parser.add_argument("--ct", action="store_true", help="use this flag for ct scans.")
parser.add_argument("--model", default=None, help="(optional) Use a different model file (Keras .h5 or .npz checkpoint).")
parser.add_argument("--disable_flipping", action="store_true",
                    help="(optional) Use this flag to disable flipping augmentation at test time.")

if __name__ == '__main__':
    args = parser.parse_args()
    print('\n\nSynthSR prediction\n\n')
    if args.cpu:
        raise NotImplementedError('--cpu: the MI355X build runs the U-Net through its HIP kernels only')
    from synthsr_amd.predict import predict
    if args.model is not None:
        print('Using user-specified model: ' + args.model)
    predict(args.path_images, args.path_predictions, path_model=args.model, ct=args.ct,
            disable_flipping=args.disable_flipping)
    print(' ')
