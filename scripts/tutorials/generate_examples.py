#!/usr/bin/env python
"""Generation examples with `BrainGenerator`, the use cases of the reference's scripts/tutorials 1-6 in one command:

  sr          super-resolution: 1 synthetic low-resolution channel in, the same contrast at the label maps' resolution out
  synthesis   contrast synthesis: channel 1 (e.g. T2-like priors) in, channel 0 (T1-like priors) out, both high resolution
  multimodal  SR + synthesis: two low-resolution channels in (different slice directions), a third contrast out
  real        real scans (--images) as regression targets for synthetic low-resolution inputs

    python scripts/tutorials/generate_examples.py sr --labels data/labels --priors data/labels_classes_priors \\
           --out generated/sr -n 3 [--shape 128]

--priors is a folder with generation_labels.npy, generation_classes.npy and prior_{means,stds}_<contrast>.npy files as
shipped with the reference (contrasts t1_hr, t1_lr, t2); each example is written as image_<i>.nii.gz (input channels,
+ reliability maps) and target_<i>.nii.gz."""
import os
import sys
import time
from argparse import ArgumentParser

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from synthsr_amd import volumes  # noqa: E402
from synthsr_amd.brain_generator import BrainGenerator  # noqa: E402

# per use case: contrasts of the channels, which are inputs, which is the target, acquisition resolution per channel
CASES = {
    'sr': dict(contrasts=['t1_lr'], input_channels=[True], output_channel=0, data_res=[[1., 1., 6.]]),
    'synthesis': dict(contrasts=['t1_hr', 't2'], input_channels=[False, True], output_channel=0, data_res=None),
    'multimodal': dict(contrasts=['t1_hr', 't1_lr', 't2'], input_channels=[False, True, True], output_channel=0,
                       data_res=[[1., 1., 3.], [1., 4.5, 1.]]),
    'real': dict(contrasts=['t1_lr'], input_channels=[True], output_channel=None, data_res=[[1., 1., 6.]]),
}


def stacked(priors_dir, kind, contrasts):
    return np.concatenate([np.load(os.path.join(priors_dir, 'prior_%s_%s.npy' % (kind, c))) for c in contrasts], axis=0)


def main(argv=None):
    ap = ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('case', choices=sorted(CASES))
    ap.add_argument('--labels', required=True, help='folder of label maps (or one label map)')
    ap.add_argument('--priors', required=True, help='folder with generation_labels/classes and prior_* arrays')
    ap.add_argument('--images', default=None, help='real scans matching the label maps (case "real")')
    ap.add_argument('--out', required=True, help='result folder')
    ap.add_argument('-n', type=int, default=3, help='number of examples')
    ap.add_argument('--shape', type=int, default=None, help='random crop to this size (multiple of 32)')
    args = ap.parse_args(argv)
    case = CASES[args.case]
    if (args.case == 'real') != (args.images is not None):
        ap.error('--images goes with the case "real" (and only with it)')
    data_res = None if case['data_res'] is None else np.array(case['data_res'])
    gen = BrainGenerator(
        labels_dir=args.labels, images_dir=args.images,
        generation_labels=os.path.join(args.priors, 'generation_labels.npy'),
        generation_classes=os.path.join(args.priors, 'generation_classes.npy'),
        prior_means=stacked(args.priors, 'means', case['contrasts']), prior_stds=stacked(args.priors, 'stds', case['contrasts']),
        prior_distributions='normal', input_channels=case['input_channels'], output_channel=case['output_channel'],
        output_shape=args.shape, output_div_by_n=32, data_res=data_res, thickness=data_res, downsample=data_res is not None,
        build_reliability_maps=data_res is not None, flipping=True, scaling_bounds=0.1, rotation_bounds=8,
        shearing_bounds=0.01, translation_bounds=False, nonlin_std=2., bias_field_std=0.2)
    os.makedirs(args.out, exist_ok=True)
    for i in range(args.n):
        t0 = time.time()
        image, target = gen.generate_brain()
        volumes.save_volume(image, gen.aff, gen.header, os.path.join(args.out, 'image_%d.nii.gz' % i))
        volumes.save_volume(target, gen.aff, gen.header, os.path.join(args.out, 'target_%d.nii.gz' % i))
        print('example %d: image %s, target %s, %.2f s' % (i, image.shape, target.shape, time.time() - t0))


if __name__ == '__main__':
    main()
