#!/usr/bin/env python
"""Inference timing of the U-Net forward used by scripts/predict_command_line.py (random weights, synthetic volume):
    python tools/predict_bench.py [size ...]     -> ms per forward pass (one flip) and per prediction with flip averaging"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from synthsr_amd.predict import Predictor
from synthsr_amd.unet import unet

sizes = [int(v) for v in sys.argv[1:]] or [160, 256]
seed_net = unet(24, [32, 32, 32, 1], 5, 3, 1, feat_mult=2, nb_conv_per_level=2, batch_norm=-1, activation='elu',
                final_pred_activation='linear', seed=0)
sd = seed_net.state_dict()
del seed_net
for n in sizes:
    p = Predictor(state_dict=sd)
    S = np.random.rand(n, n, n)
    for flip in (False, True):
        p(S, flipping=flip)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            p(S, flipping=flip)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print('%d^3 flip=%s: %.1f ms per prediction (host copies included), %.2f volumes/s' % (n, flip, ms, 1e3 / ms))
    del p
    torch.cuda.empty_cache()
