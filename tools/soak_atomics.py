"""Soak test of the default (float atomics) path (finding of round 5: the "glitches" are loss-kink ties, see below): the whole-network step of tests/test_batch_gpu.py::test_batched_unet_vs_oracle
repeated N times per configuration in ONE process, every parameter gradient of every repetition compared with the deterministic
run of the same step.  Accumulation-order noise is ~1e-6 of a tensor's range (profiles/r05_batch_test_error_distribution.txt);
anything above 1e-4 with identical max-pool choices is a glitch (a race, a stale buffer) and is printed with its repetition
number -- the round-4 driver run saw ONE such event (5.2e-3 on unet_conv_downarm_0_1/kernel) that no later run reproduced.

Result (profiles/r05_soak_atomics.txt): in 3 % of the atomics runs of the B = 2, 24-feature, 3-level case ALL 32 gradient tensors
move together, always by the same amount (5.2e-3 on unet_conv_downarm_0_1/kernel), with and without split-K (option 5 = 1) and
parity-split kernels (option 7 = 0), while every forward tensor agrees with the deterministic run: one voxel's prediction sits
within float32 rounding of its target, and sign(pred - target) of the L1 loss -- d(loss)/d(pred) = +-1/N -- takes either side.
The tool now identifies such runs from d(loss)/d(pred) (tests/conftest.py: "kinks of the loss") and reports them separately.

    python tools/soak_atomics.py [N] [arithmetic]      e.g.  python tools/soak_atomics.py 300 fp32_mfma
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'tests'))

import torch  # noqa: E402


def make_run(B, feats, levels, shape, cin, fold):
    from synthsr_amd.unet import unet
    g = torch.Generator()

    def run():
        net = unet(nb_features=feats, input_shape=list(shape) + [cin], nb_levels=levels, conv_size=3, nb_labels=1, feat_mult=2,
                   nb_conv_per_level=2, final_pred_activation='linear', batch_norm=-1, activation='elu', seed=3,
                   fold_upsample=fold)
        g.manual_seed(11)
        for nm, v in net.named_parameters():
            if nm.endswith('/gamma'):
                v.copy_(torch.rand(v.shape, generator=g) + .5)
            elif nm.endswith('/beta') or nm.endswith('/bias'):
                v.copy_(torch.randn(v.shape, generator=g) * .1)
        net.repack()
        net.set_batch(B)
        x = torch.rand(B, *shape, cin, generator=g)
        if B > 1:
            x[1] *= 1.7
        target = torch.rand(B, *shape, 1, generator=g)
        xs = x.reshape(B * shape[0], shape[1], shape[2], cin).cuda()
        _, pred = net.loss_l1(xs, target.reshape(-1).cuda(), want_pred=True)
        net.test_pred = pred.clone()
        net.backward()
        return net
    return run


def snapshot(net):
    """the forward pass's tensors in the order they are produced: conv outputs, BatchNorm batch statistics, the prediction"""
    out = []
    for l, acts in enumerate(net.saved['enc']):
        out += [('x%d (input of level %d)' % (l, l), net.saved['x'][l])]
        out += [('enc%d conv%d output' % (l, k), a) for k, a in enumerate(acts)]
    for k, acts in enumerate(net.saved['dec']):
        out += [('dec%d conv%d output' % (k, j), a) for j, a in enumerate(acts)]
    out += [('BatchNorm batch statistics', net.bn_batch), ('prediction', net.test_pred)]
    return [(nm, t.detach().float().clone()) for nm, t in out]


def main():
    from synthsr_amd import _lib, ops
    from conftest import _pool_choices, _kink_state, _kink_disagreements, _ulp_of
    _lib.load()
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    if len(sys.argv) > 2:   # (the bisect runs of round 5 passed option=value pairs of the former synthsr_conv3d_set_option here)
        ops.set_conv_arithmetic(sys.argv[2])
    print('# N = %d per configuration, arithmetic %s' % (N, ops.conv_arithmetic()))
    cases = [(2, 24, 3, (16, 16, 32), 2, True), (2, 24, 3, (16, 16, 32), 2, False), (2, 24, 4, (32, 16, 16), 2, True),
             (1, 24, 5, (32, 32, 32), 2, True)]
    for case in cases[:int(os.environ.get('SOAK_CASES', len(cases)))]:
        run = make_run(*case)
        prev = ops.set_deterministic(True)
        net = run()
        det = net.grads.clone()
        det_fwd = snapshot(net)
        det_kink = _kink_state(net)
        pool = [m.clone() for m, _ in _pool_choices(net)]
        net2 = run()
        same = torch.equal(net2.grads, det)
        ops.set_deterministic(prev)
        names = [(nm, net.offsets[nm]) for nm, _, _ in net.specs]
        rng = {nm: float(net.view(nm, det).abs().max()) for nm, _ in names}
        top = max(rng.values())
        t0, worst, events, flips, kinks, kink_ulp, kink_shift = time.time(), 0.0, 0, 0, 0, 0.0, 0.0
        for i in range(N):
            n_ = run()
            if any(not torch.equal(m0, m1) for m0, (m1, _) in zip(pool, _pool_choices(n_))):
                flips += 1      # an identified max-pool tie flip: the gradients legitimately differ
                continue
            a_pr, a_dp = _kink_state(n_)
            kidx = _kink_disagreements(a_dp, det_kink[1])
            if kidx.numel():    # an identified loss-kink tie flip: same
                kinks += 1
                kink_ulp = max(kink_ulp, float(((a_pr[kidx] - det_kink[0][kidx]).abs() / _ulp_of(det_kink[0], kidx)).max()))
                kink_shift = max(kink_shift, max(float(n_.view(nm, (n_.grads - det).abs()).max()) / max(rng[nm], 1e-3 * top) for nm, _ in names))
                continue
            diff = (n_.grads - det).abs()
            bad = []
            for nm, _ in names:
                e = float(n_.view(nm, diff).max()) / max(rng[nm], 1e-3 * top)
                worst = max(worst, e)
                if e > 1e-4:
                    bad.append((nm, e))
            if bad:     # where does the forward pass leave the deterministic one?
                events += 1
                fwd = ['%s %.1e (%d values)' % (nm, float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)), int(((a - b).abs() > 1e-4 * b.abs().max()).sum()))
                       for (nm, a), (_, b) in zip(snapshot(n_), det_fwd) if float((a - b).abs().max()) > 1e-4 * float(b.abs().max())]
                print('  GLITCH repetition %d: %d gradient tensors off (worst %s %.2e); forward tensors off by > 1e-4 of their range, in '
                      'order: %s' % (i, len(bad), max(bad, key=lambda t: t[1])[0], max(t[1] for t in bad), '; '.join(fwd[:6]) or 'none'))
        print('%s: deterministic twice bit-identical %s; %d atomics runs in %.1f s: %d runs with max-pool tie flips, %d runs with '
              'loss-kink tie flips (predictions of the flipped voxels <= %.1f ulp apart; gradients moved by up to %.2e of range); the '
              'other runs: worst %.2e of range from the deterministic gradients, %d glitches'
              % (case, same, N, time.time() - t0, flips, kinks, kink_ulp, kink_shift, worst, events))


if __name__ == '__main__':
    main()
