#!/usr/bin/env python
"""BASELINE.json configs[4] (fine_tuning_with_adversary path, "mixed bf16"): time of ONE critic update (3 forward passes,
3 backward passes, the gradient-penalty pass, Keras-Adam on 134.6 M parameters) and of the critic's share of a generator
update (forward + input gradient) at a given volume size.  Prints ONE bench-format JSON line (not the headline metric:
bench.py stays on fp32 configs[1]); the dominant conv kernel of the update is priced against the MFMA peak of the dtype or
against HBM, whichever bounds it, from HIP events recorded on the launch stream inside the timed region.

    python tools/adversarial_bench.py [--size 160] [--steps 10] [--warmup 2] [--dtype bf16|f32]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from synthsr_amd import ops  # noqa: E402
from synthsr_amd.critic import Critic3D  # noqa: E402

PEAK = {'bf16': 2500.0, 'f32': 157.3}   # dense MFMA TFLOP/s (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=160)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-settle-gc', action='store_true', help='A/B: leave the Python heap unfrozen (one update in ~12 then pays a full collection)')
    args = ap.parse_args()
    S, dtype = args.size, args.dtype
    critic = Critic3D([S, S, S, 1], seed=0, dtype=dtype)
    real, fake = torch.rand(S, S, S, 1, device='cuda'), torch.rand(S, S, S, 1, device='cuda')

    def critic_update():
        critic.critic_loss_and_grads(real, fake, 0.4, 10.0)
        critic.adam_step(1e-4)

    for _ in range(args.warmup):
        critic_update()
    torch.cuda.synchronize()
    if not args.no_settle_gc:   # what fine_tuning_with_adversary.training() does after its first step (training.settle_host_gc)
        from synthsr_amd.training import settle_host_gc
        settle_host_gc()
    nprof = min(2, args.steps)
    ops.profile_start()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        if i == nprof:
            ops.profile_pause()
        critic_update()
        marks[i + 1].record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    prof = ops.profile_stop()
    step_ms = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)])
    # generator side: forward + input gradient of the frozen critic
    critic.input_gradient(fake, -0.01)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        critic.input_gradient(fake, -0.01)
    torch.cuda.synchronize()
    gen_ms = (time.perf_counter() - t0) / args.steps * 1e3

    esz = 2 if dtype == 'bf16' else 4
    agg = {}
    for kind, shape, cin, cout, s, e in prof:
        agg.setdefault((kind, shape, cin, cout), []).append(s.elapsed_time(e))
    rows = []
    for (kind, shape, cin, cout), samples in agg.items():
        vox = float(np.prod(shape))
        cin_real = 1 if (dtype == 'bf16' and cin == 8 and shape[0] == S) else cin     # the zero-padded first layer
        up = '_up_' in kind   # folded decoder conv of the generator (low-resolution shape): 8 parity convs of 2x2x2 taps
        fl = 2.0 * (64 if up else 27) * cin_real * cout * vox
        by = esz * vox * (cin_real + (8 if up else 1) * cout)
        ms, cnt = float(sum(samples)), len(samples)
        rows.append(dict(kernel=kind, shape=list(shape), cin=cin, cout=cout, launches=cnt, avg_ms=ms / cnt,
                         tflops=fl / (ms / cnt * 1e-3) / 1e12, gbs=by / (ms / cnt * 1e-3) / 1e9, total_ms=ms, flops=fl,
                         bytes=by))
    rows.sort(key=lambda r: -r['total_ms'])
    roof = None
    if rows:
        dom = rows[0]
        balance = PEAK[dtype] * 1e12 / (HBM_PEAK_GBS * 1e9)
        intensity = dom['flops'] / dom['bytes']
        if intensity < balance:
            roof = {'bound': 'hbm', 'achieved': round(dom['gbs'], 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(dom['gbs'] / HBM_PEAK_GBS, 4)}
        else:
            roof = {'bound': 'mfma', 'achieved': round(dom['tflops'], 2), 'peak': PEAK[dtype], 'unit': 'TFLOP/s',
                    'frac': round(dom['tflops'] / PEAK[dtype], 4)}
        roof.update(kernel='%s %s Cin=%d Cout=%d' % (dom['kernel'], 'x'.join(map(str, dom['shape'])), dom['cin'], dom['cout']),
                    traffic=None, traffic_source=None, avg_launch_ms=round(dom['avg_ms'], 4),
                    flop_per_byte=round(intensity, 1), machine_balance=round(balance, 1), algorithmic_bytes=dom['bytes'],
                    flops_per_launch=dom['flops'], conv_ms_per_step=round(sum(r['total_ms'] for r in rows) / nprof, 3),
                    profiled_steps=nprof)
    out = {'metric': 'critic updates/sec (%d^3 %s, WGAN-GP critic of fine_tuning_with_adversary)' % (S, dtype),
           'value': round(1.0 / dt, 3), 'unit': 'updates/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
           'ms_per_step': round(dt * 1e3, 3),
           'step_ms': {'mean': round(float(step_ms.mean()), 3), 'median': round(float(np.median(step_ms)), 3),
                       'min': round(float(step_ms.min()), 3), 'max': round(float(step_ms.max()), 3),
                       'each': [round(float(v), 2) for v in step_ms]},
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic',
           'config': {'workload': 'configs[4]: one WGAN-GP critic update (-D(real) + D(fake) + 10 (1 - |grad D(x_hat)|)^2: 3 '
                                  'forward, 3 backward, penalty pass, Adam) at %d^3, %.1f M parameters (Dense %d x %d), %s'
                                  % (S, critic.n_params / 1e6, critic.dense[0]['n_in'], critic.dense[0]['n_out'],
                                     'bf16 conv stack, fp32 accumulation / Dense / master weights' if dtype == 'bf16'
                                     else 'fp32'),
                      'global_batch': 1, 'parallelism': 'dp1', 'volume': [S, S, S]},
           'generator_update_critic_share_ms': round(gen_ms, 3), 'roofline': roof,
           'top_kernels': [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()
                            if k not in ('flops', 'bytes')} for r in rows[:8]]}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
