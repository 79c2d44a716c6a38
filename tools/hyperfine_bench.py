#!/usr/bin/env python
"""BASELINE.json configs[3]: Hyperfine-like two-channel generation (input_channels=[False, True, True], 1.5 x 1.5 x 5 mm
acquisitions, registration error, no reliability maps) at 192^3 + the 5-level U-Net (Cin = 2) with the residual on the
first input channel + Adam, in bf16 (bf16 activations / weights, fp32 accumulation, fp32 BatchNorm statistics, fp32 master
weights) or fp32.  Prints ONE bench-format JSON line (not the headline metric: bench.py stays on fp32 configs[1]).

    python tools/hyperfine_bench.py [--dtype bf16|f32] [--steps K] [--warmup W] [--size S] [--config hyperfine|c1]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from synthsr_amd import ops  # noqa: E402
from synthsr_amd.brain_generator import BrainGenerator  # noqa: E402
from synthsr_amd.synthetic import (synthetic_label_pool, GENERATION_LABELS, GENERATION_CLASSES, PRIOR_MEANS_T1_HR,  # noqa: E402
                                   PRIOR_STDS_T1_HR)
from synthsr_amd.training import Trainer  # noqa: E402
from synthsr_amd.unet import unet  # noqa: E402

PEAK = {'bf16': 2500.0, 'f32': 157.3}   # dense MFMA TFLOP/s (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--size', type=int, default=192)
    ap.add_argument('--config', default='hyperfine', choices=['hyperfine', 'c1'])
    args = ap.parse_args()
    S = args.size
    pool = synthetic_label_pool(4, (S, S, S), 1234)
    if args.config == 'hyperfine':
        res = np.array([[1.5, 1.5, 5.], [1.5, 1.5, 5.]])
        pm = np.concatenate([PRIOR_MEANS_T1_HR] * 3)    # three contrasts (tutorial 6 concatenates t1_hr, t1_lr, t2 priors)
        ps = np.concatenate([PRIOR_STDS_T1_HR] * 3)
        bg = BrainGenerator(None, pm, ps, 'normal', GENERATION_LABELS, generation_classes=GENERATION_CLASSES,
                            n_neutral_labels=19, input_channels=[False, True, True], output_channel=0, output_shape=S,
                            output_div_by_n=32, data_res=res, thickness=res, downsample=True, build_reliability_maps=False,
                            simulate_registration_error=True, blur_range=1.15, nonlin_shape_factor=.03125,
                            bias_shape_factor=.03125, label_maps=pool, rng=np.random.Generator(np.random.Philox(key=7)))
        residual = [0]
    else:
        bg = BrainGenerator(None, PRIOR_MEANS_T1_HR, PRIOR_STDS_T1_HR, 'normal', GENERATION_LABELS,
                            generation_classes=GENERATION_CLASSES, n_neutral_labels=19, output_shape=S, output_div_by_n=32,
                            nonlin_shape_factor=.03125, bias_shape_factor=.03125, build_reliability_maps=True,
                            downsample=True, shearing_bounds=.02, translation_bounds=5, label_maps=pool,
                            rng=np.random.Generator(np.random.Philox(key=7)))
        residual = None
    bg.labels_to_image_model.seed(0, 0)
    net = unet(24, bg.model_output_shape, 5, 3, 1, feat_mult=2, nb_conv_per_level=2, final_pred_activation='linear',
               batch_norm=-1, activation='elu', seed=0, dtype=args.dtype)
    tr = Trainer(bg, net, lr=1e-4, work_with_residual_channel=residual)
    tr.make_labels_resident(pool)
    pick = np.random.default_rng(0)
    for _ in range(args.warmup):
        loss = tr.step(label_index=int(pick.integers(len(pool))))
    torch.cuda.synchronize()
    ops.profile_start()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == min(3, args.steps):
            ops.profile_pause()
        loss = tr.step(label_index=int(pick.integers(len(pool))))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    prof = ops.profile_stop()
    agg = {}
    for kind, shape, cin, cout, s, e in prof:
        if not kind.startswith('gen'):
            agg.setdefault((kind, shape, cin, cout), []).append(s.elapsed_time(e))
    rows = []
    for (kind, shape, cin, cout), samples in agg.items():
        up = '_up_' in kind                                    # folded decoder conv, recorded with its LOW-resolution shape:
        taps = 64 if up else 27                                # 8 parity convs of 2x2x2 taps per low-resolution voxel
        cin_real = min(cin, 2) if cin == 8 else cin            # the zero-padded first layer: algorithmic channels
        fl_alg = 2.0 * taps * cin_real * cout * float(np.prod(shape))
        esz = 2 if args.dtype == 'bf16' else 4
        by = esz * float(np.prod(shape)) * (cin_real + (8 if up else 1) * cout)
        ms, cnt = float(sum(samples)), len(samples)
        rows.append(dict(kernel=kind, shape=list(shape), cin=cin, cout=cout, launches=cnt, avg_ms=ms / cnt,
                         tflops=fl_alg / (ms / cnt * 1e-3) / 1e12, gbs=by / (ms / cnt * 1e-3) / 1e9, total_ms=ms,
                         flops=fl_alg, bytes=by))
    rows.sort(key=lambda r: -r['total_ms'])
    dom = rows[0]
    conv_total = sum(r['total_ms'] for r in rows)
    nprof = min(3, args.steps)
    # which roofline binds the dominant kernel: arithmetic intensity vs the machine balance of this dtype
    balance = PEAK[args.dtype] * 1e12 / (HBM_PEAK_GBS * 1e9)
    intensity = dom['flops'] / dom['bytes']
    if intensity < balance:
        roof = {'bound': 'hbm', 'achieved': round(dom['gbs'], 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(dom['gbs'] / HBM_PEAK_GBS, 4)}
    else:
        roof = {'bound': 'mfma', 'achieved': round(dom['tflops'], 2), 'peak': PEAK[args.dtype], 'unit': 'TFLOP/s',
                'frac': round(dom['tflops'] / PEAK[args.dtype], 4)}
    roof.update(kernel='%s %s Cin=%d Cout=%d' % (dom['kernel'], 'x'.join(map(str, dom['shape'])), dom['cin'], dom['cout']),
                traffic=None, avg_launch_ms=round(dom['avg_ms'], 4), flop_per_byte=round(intensity, 1),
                machine_balance=round(balance, 1), algorithmic_bytes=dom['bytes'], flops_per_launch=dom['flops'],
                mfma_tflops=round(dom['tflops'], 1), conv_ms_per_step=round(conv_total / nprof, 3))
    name = 'configs[3]: Hyperfine-like [False, True, True] 1.5x1.5x5 mm, registration error' if args.config == 'hyperfine' \
        else 'configs[1] generator'
    out = {'metric': 'training volumes/sec (%d^3 %s, 5-level U-Net)' % (S, args.dtype), 'value': round(1.0 / dt, 3),
           'unit': 'volumes/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt * 1e3, 3),
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
           'config': {'workload': '%s, %d^3 + 5-level 3-D U-Net (24..384 features, Cin=2, residual %s) fwd/bwd + Adam, %s'
                                  % (name, S, residual, 'bf16 activations/weights, fp32 accumulation + BatchNorm statistics '
                                     '+ master weights' if args.dtype == 'bf16' else 'fp32'),
                      'global_batch': 1, 'parallelism': 'dp1', 'volume': [S, S, S]},
           'roofline': roof, 'final_loss': round(float(loss.item()), 6),
           'top_kernels': [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()
                            if k not in ('flops', 'bytes')} for r in rows[:24]]}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
