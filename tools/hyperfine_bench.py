#!/usr/bin/env python
"""Step time of BASELINE.json configs[3] in fp32 (the config names bf16; this build computes in fp32): Hyperfine-like
two-channel generation (input_channels=[False, True, True], 1.5 x 1.5 x 5 mm acquisitions, registration error, no
reliability maps) at 192^3 + the 5-level U-Net (Cin = 2) with the residual on the first input channel + Adam.

    python tools/hyperfine_bench.py [steps] [size]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from synthsr_amd.brain_generator import BrainGenerator  # noqa: E402
from synthsr_amd.synthetic import (synthetic_label_pool, GENERATION_LABELS, GENERATION_CLASSES, PRIOR_MEANS_T1_HR,  # noqa: E402
                                   PRIOR_STDS_T1_HR)
from synthsr_amd.training import Trainer  # noqa: E402
from synthsr_amd.unet import unet  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 192
    pool = synthetic_label_pool(4, (S, S, S), 1234)
    res = np.array([[1.5, 1.5, 5.], [1.5, 1.5, 5.]])
    pm = np.concatenate([PRIOR_MEANS_T1_HR] * 3)        # three contrasts (tutorial 6 concatenates t1_hr, t1_lr, t2 priors)
    ps = np.concatenate([PRIOR_STDS_T1_HR] * 3)
    bg = BrainGenerator(None, pm, ps, 'normal', GENERATION_LABELS, generation_classes=GENERATION_CLASSES, n_neutral_labels=19,
                        input_channels=[False, True, True], output_channel=0, output_shape=S, output_div_by_n=32,
                        data_res=res, thickness=res, downsample=True, build_reliability_maps=False,
                        simulate_registration_error=True, blur_range=1.15, nonlin_shape_factor=.03125,
                        bias_shape_factor=.03125, label_maps=pool, rng=np.random.Generator(np.random.Philox(key=7)))
    bg.labels_to_image_model.seed(0, 0)
    net = unet(24, bg.model_output_shape, 5, 3, 1, feat_mult=2, nb_conv_per_level=2, final_pred_activation='linear',
               batch_norm=-1, activation='elu', seed=0)
    tr = Trainer(bg, net, lr=1e-4, work_with_residual_channel=[0])
    for _ in range(3):
        loss = tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print('configs[3] in fp32: %d^3, U-Net input %s: %.2f ms per generate + train step = %.2f volumes/s (loss %.4f)'
          % (S, bg.model_output_shape, dt * 1e3, 1 / dt, float(loss.item())))


if __name__ == '__main__':
    main()
