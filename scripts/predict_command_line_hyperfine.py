#!/usr/bin/env python
"""MI355X build of the reference's scripts/predict_command_line_hyperfine.py (same positional arguments and flags):
    python scripts/predict_command_line_hyperfine.py <path_t1_images> <path_t2_images> <path_predictions>
--cpu / --threads are accepted for command-line compatibility; this build has no CPU inference path."""
import os
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

parser = ArgumentParser()
parser.add_argument("path_t1_images", type=str, help="T1 images to super-resolve, at native 1.5x1.5x5 axial resolution. "
                                                     "Can be the path to a single image or to a folder")
parser.add_argument("path_t2_images", type=str, help="T2 images to super-resolve, at native 1.5x1.5x5 axial resolution. "
                                                     "Can be the path to a single image or to a folder")
parser.add_argument("path_predictions", type=str, help="path where to save the synthetic 1mm MP-RAGEs. Must be the same "
                                                       "type as path_images (path to a single image or to a folder)")
parser.add_argument("--cpu", action="store_true", help="(reference flag) CPU inference: not available in this build.")
parser.add_argument("--threads", type=int, default=1, dest="threads", help="(reference flag) ignored.")
parser.add_argument("--model", default=None, help="(optional) Use a different model file (Keras .h5 or .npz checkpoint).")

if __name__ == '__main__':
    args = parser.parse_args()
    print('\n\nSynthSR-hyperfine prediction\n\n')
    if args.cpu:
        raise NotImplementedError('--cpu: the MI355X build runs the U-Net through its HIP kernels only')
    from synthsr_amd.predict import predict_hyperfine
    predict_hyperfine(args.path_t1_images, args.path_t2_images, args.path_predictions, path_model=args.model)
    print(' \nAll done!\n ')
