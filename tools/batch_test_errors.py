"""Error distribution of the whole-network batch test that went red on the driver's box in round 4
(tests/test_batch_gpu.py::test_batched_unet_vs_oracle[2-24-3-shape0-2-True]).

For every kernel variant below, N repetitions of the test's own step (same seeds: the deterministic run is identical every
time, the atomics run differs by its accumulation order) and N steps with a different input seed each.  Reported per run for
`unet_conv_downarm_0_1/kernel` (the tensor that failed) and as the worst over all tensors, each as max |error| / max |reference|:

  det/f32   deterministic device run vs the fp32 oracle          det/f64   ... vs a FLOAT64 run of the oracle
  atm/f32   atomics device run vs the fp32 oracle                atm/f64   ... vs float64
  atm-det   atomics run vs deterministic run                     o32/f64   the fp32 oracle itself vs float64

Variants: split (default) and fp32_mfma arithmetic.  (The run recorded in profiles/r05_batch_test_error_distribution.txt, made
before the process-wide option switch was removed, also covers option 12=9 -- 8 instead of all 24 input channels per stacked
24-column workgroup, the round-4 23:43 kernel -- and option 11=256 -- layers below 256 tiles on the fp32-MFMA weight gradient,
the round-3 rule.)  Max-pool rounding ties between device and oracle are aligned as in tests/conftest.py.

    python tools/batch_test_errors.py [N] > profiles/r05_batch_test_error_distribution.txt
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'tests'))

import torch  # noqa: E402


def step_errors(B, feats, levels, shape, cin, fold, seed, focus):
    from synthsr_amd import ops
    from synthsr_amd.unet import unet
    from oracle import unet_ref as U
    from conftest import _pool_choices, align_pool_ties
    g = torch.Generator()
    tensors = {}

    def run():
        net = unet(nb_features=feats, input_shape=list(shape) + [cin], nb_levels=levels, conv_size=3, nb_labels=1, feat_mult=2,
                   nb_conv_per_level=2, final_pred_activation='linear', batch_norm=-1, activation='elu', seed=3,
                   fold_upsample=fold)
        g.manual_seed(seed)
        for nm, v in net.named_parameters():
            if nm.endswith('/gamma'):
                v.copy_(torch.rand(v.shape, generator=g) + .5)
            elif nm.endswith('/beta') or nm.endswith('/bias'):
                v.copy_(torch.randn(v.shape, generator=g) * .1)
        net.repack()
        net.set_batch(B)
        x = torch.rand(B, *shape, cin, generator=g)
        x[1] *= 1.7
        target = torch.rand(B, *shape, 1, generator=g)
        xs = x.reshape(B * shape[0], shape[1], shape[2], cin).cuda()
        net.loss_l1(xs, target.reshape(-1).cuda(), want_pred=True)
        net.backward()
        tensors.update(x=x, target=target)
        return net

    def oracle(net, pool, dtype):
        def once(nudge):
            P = {nm: v.detach().cpu().clone().requires_grad_(True) for nm, v in net.named_parameters()}
            pin = []
            with U.compute_dtype(dtype):
                pr = U.unet_forward(tensors['x'], P, net.prefix, levels, 2, training=True, pool_inputs=pin, pool_nudge=nudge)
                U.l1_loss(pr, tensors['target']).backward()
            return P, pin
        P, pin = once(None)
        nudges, n = align_pool_ties(pool, pin, max_ties=64)
        if n:
            P, _ = once(nudges)
        return {k: v.grad.double() for k, v in P.items()}, n

    prev = ops.set_deterministic(True)
    try:
        net = run()
        pool = list(_pool_choices(net))
        det = {nm: net.view(nm, net.grads).cpu().double() for nm, _, _ in net.specs}
    finally:
        ops.set_deterministic(prev)
    g32, t32 = oracle(net, pool, None)
    g64, t64 = oracle(net, pool, torch.float64)
    net = run()
    atm = {nm: net.view(nm, net.grads).cpu().double() for nm, _, _ in net.specs}
    flips = sum(int((m0 != m1).sum()) for (m0, _), (m1, _) in zip(pool, _pool_choices(net)))

    def e(a, b, nm):
        return float((a[nm] - b[nm]).abs().max() / b[nm].abs().max().clamp_min(1e-30))
    rows = {}
    for nm in det:
        rows[nm] = (e(det, g32, nm), e(det, g64, nm), e(atm, g32, nm), e(atm, g64, nm), e(atm, det, nm), e(g32, g64, nm))
    worst = tuple(max(r[k] for r in rows.values()) for k in range(6))
    return rows[focus], worst, (t32, t64, flips), rows


def main():
    from synthsr_amd import _lib, ops
    _lib.load()
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    focus = 'unet_conv_downarm_0_1/kernel'
    case = dict(B=2, feats=24, levels=3, shape=(16, 16, 32), cin=2, fold=True)
    variants = [('default', [], 'split'), ('fp32_mfma arithmetic', [], 'fp32_mfma')]
    print('# %s, N = %d; columns: det/f32 det/f64 atm/f32 atm/f64 atm-det o32/f64 | worst tensor: same six | pool ties (f32, f64 '
          'oracle), det->atm flips' % (case, N))
    for name, opts, arith in variants:
        ops.set_conv_arithmetic(arith)
        print('\n## %s' % name)
        for label, seeds in (('test seed 11, repeated', [11] * N), ('input seeds 100..', list(range(100, 100 + N)))):
            print('# %s' % label)
            acc = []
            for s in seeds:
                f, w, ties, rows = step_errors(focus=focus, seed=s, **case)
                acc.append(f + w)
                print('  ' + ' '.join('%.2e' % v for v in f) + ' | ' + ' '.join('%.2e' % v for v in w) + ' | %d %d %d' % ties)
            t = torch.tensor(acc)
            print('  median ' + ' '.join('%.2e' % v for v in t.median(0).values.tolist()))
            print('  max    ' + ' '.join('%.2e' % v for v in t.max(0).values.tolist()))
        if name == 'default':
            print('# every tensor of the last step (det/f32 det/f64 atm/f32 atm/f64 atm-det o32/f64):')
            for nm, r in rows.items():
                print('  %-36s ' % nm + ' '.join('%.2e' % v for v in r))
    ops.set_conv_arithmetic('split')


if __name__ == '__main__':
    main()
