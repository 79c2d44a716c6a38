import os
import sys
import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def tape_from_golden(g, prefix='tape'):
    kinds = g['%s_kinds' % prefix]
    return [(str(k), g['%s_%02d' % (prefix, i)]) for i, k in enumerate(kinds)]


@pytest.fixture(scope='session')
def gen_labels():
    return np.array([0, 14, 15, 16, 2, 3, 4, 5, 7, 8, 10, 11, 12, 13, 17, 18, 26, 28, 31], dtype=np.int32)


def regen_weights(names, shapes, seed):
    """the seeded weight recipe of tests/golden/gen/keras_layers_shim._weight, replayed in the order of `names`
    (used for goldens whose 13 M weights are not stored; the golden holds per-tensor checksums)"""
    rng = np.random.default_rng(int(seed))
    out = {}
    for nm, shp in zip(names, shapes):
        nm = str(nm)
        shape = tuple(int(s) for s in shp if s > 0)
        kind = nm.split('/')[-1]
        if kind == 'kernel':
            rf = int(np.prod(shape[:-2]))
            lim = np.sqrt(6.0 / (rf * shape[-2] + rf * shape[-1]))
            a = rng.uniform(-lim, lim, shape)
        elif kind == 'bias':
            a = rng.normal(0, .05, shape)
        elif kind == 'gamma':
            a = rng.uniform(.5, 1.5, shape)
        elif kind == 'beta':
            a = rng.normal(0, .1, shape)
        elif kind == 'moving_mean':
            a = rng.normal(0, .2, shape)
        elif kind == 'moving_variance':
            a = rng.uniform(.5, 2., shape)
        else:
            raise KeyError(nm)
        out[nm] = a.astype(np.float32)
    return out


def golden_weights(g, prefix):
    """all arrays stored under '<prefix><layer>/<weight>' of a golden file"""
    return {k[len(prefix):]: g[k] for k in g.files if k.startswith(prefix)}


def random_tape(m, rng):
    """a random tape (list of (kind, array)) for the generator model `m`, in the reference's call order -- the mirror of
    LabelsToImageModel.draws_from_tape / oracle.generator_ref.labels_to_image's reads"""
    u = lambda k: ('u', rng.random(k, dtype=np.float32))
    n = lambda k: ('n', rng.standard_normal(k, dtype=np.float32))
    t = []
    if m.rotation_bounds is not False:
        t.append(u(3))
    if m.shearing_bounds is not False:
        t.append(u(6))
    if m.scaling_bounds is not False:
        t.append(u(3))
    if m.translation_bounds is not False:
        t.append(u(3))
    if m.apply_elastic:
        t += [u(1), n(int(np.prod(m.small_shape)) * 3)]
    if m.crop_shape != m.labels_shape:
        t.append(u(3))
    if m.flipping:
        t.append(u(1))
    t.append(n(m.ncrop * m.n_channels))
    for i in range(m.n_channels):
        if m.input_channels[i] and m.bias_field_std > 0:
            t += [u(1), n(int(np.prod(m.small_bias_shape))), u(1)]
        t.append(n(1))
        if m.input_channels[i]:
            reg = bool(m.simulate_registration_error[i]) and i != m.idx_first_input_channel
            if reg:
                t += [u(3), u(3)]
            if m.randomise_res[i]:
                t += [u(1), u(3), u(1), u(3)]
            if m.blur_range is not None and m.blur_range != 1:
                t.append(u(3))
            if reg:
                t += [u(3), u(3)]
    return t


def tape_from_draws(m, d, gmm_noise):
    """the tape (reference call order) equivalent to a Draws object `d` of model `m`; `gmm_noise` [ncrop * C] stands in
    for the in-kernel Philox stream (oracle.philox_ref.normals of d.philox_key / d.philox_offset)"""
    f = lambda a: np.asarray(a, dtype=np.float32).reshape(-1)
    t = []
    for kind, v in (('u', d.u_rot), ('u', d.u_shear), ('u', d.u_scale), ('u', d.u_trans)):
        if v is not None:
            t.append((kind, f(v)))
    if m.apply_elastic:
        t += [('u', f(d.u_svf_std)), ('n', f(d.n_svf))]
    if m.crop_shape != m.labels_shape:
        t.append(('u', f(d.u_crop)))
    if m.flipping:
        t.append(('u', f(d.u_flip)))
    t.append(('n', f(gmm_noise)))
    for i in range(m.n_channels):
        c = d.channels[i]
        if 'u_bias_std' in c:
            t += [('u', f(c['u_bias_std'])), ('n', f(c['n_bias'])), ('u', f(c['u_bias_gate']))]
        t.append(('n', f(c['n_gamma'])))
        if 'u_regT' in c:
            t += [('u', f(c['u_regT'][0])), ('u', f(c['u_regT'][1]))]
        if 'u_rr' in c:
            t += [('u', f(v)) for v in c['u_rr']]
        if 'u_blur' in c:
            t.append(('u', f(c['u_blur'])))
        if 'u_regE' in c:
            t += [('u', f(c['u_regE'][0])), ('u', f(c['u_regE'][1]))]
    return t


def _pool_choices(net):
    """the device's OWN arg-max choice of every 2x2x2 max-pool of the step in flight: bn_maxpool_bwd routes a gradient of ones
    to the winner of each window, so the non-zeros of its output ARE the arg-max mask the backward pass used.  Returns
    [(mask bool [d0,d1,d2,C], BatchNorm output float32 recomputed in torch)] per pooled level."""
    import torch
    from synthsr_amd import ops
    out = []
    for l in range(net.nb_levels - 1):
        e = net.enc[l]
        # per-sample dropout (batchsize > 1): the BatchNorm + pooling read the dropped-out copy of the conv output
        x = (net.saved['encd'] if getattr(net, '_drop_ps', None) is not None else net.saved['enc'])[l][-1]
        st, C = net._stats(e['bn']), e['bn']['C']
        gamma, beta = net.view(e['bn']['gamma']), net.view(e['bn']['beta'])
        ones = torch.ones([x.shape[0] // 2, x.shape[1] // 2, x.shape[2] // 2, C], dtype=x.dtype, device=x.device)
        mask = ops.bn_maxpool_bwd(ones, x, st, gamma, beta).float() != 0
        inv = torch.rsqrt(st[C:2 * C] + ops.BN_EPS) * gamma
        out.append((mask, x.float() * inv + (beta - st[:C] * inv)))
    return out


def _windows(t):
    d0, d1, d2, C = t.shape
    return t.reshape(d0 // 2, 2, d1 // 2, 2, d2 // 2, 2, C).permute(0, 2, 4, 6, 1, 3, 5).reshape(-1, 8)


def _unwindows(w, shape):
    d0, d1, d2, C = shape
    return w.reshape(d0 // 2, d1 // 2, d2 // 2, C, 2, 2, 2).permute(0, 4, 1, 5, 2, 6, 3).reshape(d0, d1, d2, C)


def align_pool_ties(dev_choices, oracle_inputs, max_ties=8, max_ulp=4.0, report=None):
    """Max-pooling is discontinuous: when two candidates of a 2x2x2 window are within float32 rounding of each other, WHICH
    one wins depends on the last bits of the BatchNorm statistics, i.e. on the summation order of the implementation -- the
    device and the CPU oracle may legitimately differ there, and the level's gradients then differ by a few 1e-3 of their
    range although both are correct.  This function compares the device's own arg-max choices (`_pool_choices`) with the
    oracle's pre-pooling tensors (oracle.unet_ref.unet_forward(pool_inputs=...)), window by window.  Every disagreement must
    be a TIE -- the oracle's value at the device's choice within 4 ulp of the oracle's maximum, an ulp being taken of the larger
    of the two candidates and the tensor's rms (the candidates are BatchNorm outputs (x - mean) / std: their rounding error is
    set by the magnitude of x and mean, i.e. by the tensor's scale, not by how close to zero the difference lands) -- and there may be at most
    `max_ties`; anything else raises (`max_ulp`: the full-size test widens the 4 ulp to the measured distance between the two
    implementations' BatchNorm statistics over 4 M voxels).  Returns (nudges, n_ties): per pooled level None or a tensor (oracle layout) that, added
    before the oracle's pooling (pool_nudge=...), makes it break exactly those ties the way the device did.
    report (optional dict) receives 'windows' (number of pooling windows compared) and 'ulps' (the distance of every aligned
    window's two candidates, in ulp as defined above, one tensor per level)."""
    import torch
    eps = torch.finfo(torch.float32).eps
    nudges, n_ties = [], 0
    for l, ((mask, tdev), t) in enumerate(zip(dev_choices, oracle_inputs)):
        oshape = t.shape
        t = t.reshape(-1, *t.shape[-3:]) if t.dim() == 5 else t          # a batch [B, d0, ...] -> the stack [B d0, ...]
        w = _windows(t.float())
        level_ulp = max_ulp
        if tdev is not None and tuple(tdev.shape) == tuple(t.shape):
            # (round 6) the two implementations' values of THIS tensor are both at hand: candidates closer than twice the distance
            # between them cannot be ordered by either -- a few ulp in the network under test, more in a network downstream of
            # its prediction (the frozen segmentation net inherits the prediction's 20-40 ulp); capped at 64 ulp
            rms_t = max(float(w.pow(2).mean().sqrt()), 1e-30)
            apart = float((tdev.detach().float().cpu() - t.float()).abs().max()) / (eps * rms_t)
            level_ulp = min(64.0, max(max_ulp, 2.0 * apart))
        if report is not None:
            report['windows'] = report.get('windows', 0) + int(w.shape[0])
        idx_dev = _windows(mask.cpu()).float().argmax(1)
        idx_or = w.argmax(1)                                              # first maximum in raster order, like max_pool3d
        diff = (idx_dev != idx_or).nonzero().reshape(-1)
        if diff.numel() == 0:
            nudges.append(None)
            continue
        a, b = w[diff, idx_dev[diff]], w[diff, idx_or[diff]]
        ulp = eps * torch.maximum(a.abs(), b.abs()).clamp_min(float(w.pow(2).mean().sqrt()))
        worst = float(((b - a) / ulp).max())
        if report is not None:
            report.setdefault('ulps', []).append(((b - a) / ulp).clone())
        assert worst <= level_ulp, 'pooled level %d: device and oracle pick different maxima in %d windows whose candidates are up ' \
            'to %.1f ulp apart (allowed %.1f): not a rounding tie' % (l, diff.numel(), worst, level_ulp)
        nud = torch.zeros_like(w)
        nud[diff, idx_dev[diff]] = (b - a) + 8 * ulp
        nudges.append(_unwindows(nud, t.shape).reshape(oshape))
        n_ties += int(diff.numel())
    assert n_ties <= max_ties, '%d pooling ties between device and oracle (> %d)' % (n_ties, max_ties)
    return (nudges if n_ties else None), n_ties


# ---- gradient bounds anchored on float64 --------------------------------------------------------------
# A whole-network gradient is a long chain of fp32 roundings pushed through BatchNorm layers over a few hundred values: how far
# ANY correct fp32 evaluation lands from the exact result depends on the network, the tensor and the data, so a fixed bound is
# either slack or a coin.  The yardstick is the oracle itself: evaluated once in float32 and once in float64
# (oracle.unet_ref.compute_dtype), per tensor
#     d = max |device - float64| / range        o = max |fp32 oracle - float64| / range
# and the device must satisfy  d <= GRAD_K * o + GRAD_FLOOR  -- never more than GRAD_K times further from the truth than the
# fp32 host evaluation of the same graph, plus a floor for tensors the host happens to hit almost exactly (the rule of the
# 160^3 step, tests/test_full_size_parity_gpu.py).  range = max |float64 gradient| of the tensor, but at least RANGE_FLOOR of
# the largest gradient of the network (a bias whose gradient is a sum of cancelling terms has no range of its own).  The floor
# is 2e-5 for conv / head kernels and 4x that for the per-channel SUMS over every voxel (biases, BatchNorm beta / gamma): their
# terms cancel, and the rounding of a sum scales with the sum of |terms|, not with |sum| = the range.
# Measured over the 47 whole-network tests of the suite at the time (3016 tensor records; 48 tests since, profiles/r05_parity_anchor_distribution.txt): device
# vs float64 at most 2.5e-5 (kernels) / 3.0e-5 (sums) of range -- median 1.6e-6 --, the fp32 oracle at most 1.6e-5 / 2.8e-5;
# device / oracle 1.2 in the median, 3.2-4.2 at the 99th percentile; the tightest record sits 2.6x inside the rule.
GRAD_K, GRAD_FLOOR, RANGE_FLOOR = 6.0, 2e-5, 1e-3
SUM_FLOOR_FACTOR = 4.0


def _anchor_log(rows, tag):
    out_dir = os.path.join(REPO, 'gpurun_out')
    if os.path.isdir(out_dir):   # scratch record of every (d, o) pair: profiles/r05_parity_anchor_distribution.txt is made of it
        import json
        with open(os.path.join(out_dir, 'parity_anchor.jsonl'), 'a') as f:
            f.write(json.dumps({'test': os.environ.get('PYTEST_CURRENT_TEST', ''), 'tag': tag, 'rows': rows}) + '\n')


def assert_grads_anchored(dev, g32, g64, kinds=None, tag='', k=GRAD_K, floor=GRAD_FLOOR, extra=None):
    """dev, g32, g64: name -> gradient (device / fp32 oracle / float64 oracle).  extra (optional): name -> another device
    result (the atomics run) that must lie within k * o + floor of `dev`.  Returns {name: (d, o)}."""
    top = max(float(g64[nm].abs().max()) for nm in dev)
    rows, bad = [], []
    for nm in dev:
        ref = g64[nm].double().cpu()
        rng = max(float(ref.abs().max()), RANGE_FLOOR * top, 1e-300)
        d = float((dev[nm].double().cpu() - ref).abs().max()) / rng
        o = float((g32[nm].double().cpu() - ref).abs().max()) / rng
        a = None if extra is None else float((extra[nm].double().cpu() - dev[nm].double().cpu()).abs().max()) / rng
        rows.append((nm, d, o, a))
        fl = floor * (1.0 if nm.endswith('/kernel') else SUM_FLOOR_FACTOR)
        if d > k * o + fl or (a is not None and a > k * o + fl):
            bad.append((nm, d, o, a))
    _anchor_log(rows, tag)
    assert not bad, '%s: gradients further from float64 than %g x the fp32 oracle + %g (x %g for per-channel sums) (name, device, fp32 oracle%s): %s' % (
        tag, k, floor, SUM_FLOOR_FACTOR, ', second run vs first' if extra is not None else '', bad)
    return {nm: (d, o) for nm, d, o, _ in rows}


# ---- kinks of the loss at the prediction level -------------------------------------------------------------------------------
# |pred - target| (L1, the Laplace likelihood) and the clip of the segmentation-regularised loss are not differentiable where
# their argument is 0 / on the clip bound: a voxel whose prediction is within float32 rounding of its target gets d(loss)/d(pred)
# = +1/N from one correct implementation and -1/N from another, and ONE such voxel moves every parameter gradient of a 16 k-voxel
# test network by up to 2 / sqrt(N) ~ 1 % of its size.  tools/soak_atomics.py (profiles/r05_soak_atomics.txt) found exactly this
# behind the round-4 red test: the atomics run of test_batched_unet_vs_oracle[2-24-3-shape0-2-*] lands on the other side of one
# voxel's kink in 3 % of its runs (all 32 gradient tensors move together, always by the same 5.2e-3; the forward pass agrees to
# the last bits).  Same treatment as max-pool ties: the disagreeing voxels are IDENTIFIED from the two d(loss)/d(pred) tensors,
# each must be a rounding tie (the two predictions a few ulp apart with the kink between them), there may be only a handful, and
# the oracle is re-run with its prediction moved onto the device's side of the kink at exactly those voxels.
# How far apart two correct predictions of the SAME voxel may be for a kink between them to count as a rounding tie: 128 ulp (an ulp
# of the larger of |pred| and the tensor's rms) = 1.5e-5 of the prediction's scale.  Measured on these 16 k-voxel networks, whose
# BatchNorm statistics run over a few hundred values: device vs fp32 oracle up to 29 ulp, vs the float64 oracle 23, atomics run vs
# deterministic run 18 (soak, profiles/r05_soak_atomics.txt) to 36 (test_loss_kink_ties_are_identified_and_aligned).
KINK_ULP = 128.0


def _kink_state(net):
    """(prediction or None, d(loss)/d(prediction)) of the step in flight, flat [voxel][head channel], float32 on the host"""
    dp = net.dpred.detach().float().cpu().clone()
    pr = net._bufs.get('pred')
    return (None if pr is None else pr[:dp.numel()].detach().float().cpu().clone()), dp


def _kink_disagreements(dp_a, dp_b):
    """entries whose loss derivative differs grossly between two evaluations: more than 1e-3 of the largest |d(loss)/d(pred)|
    (rounding differences are ~1e-6 of it; a sign flip of an L1 term is 2x it)"""
    return ((dp_a - dp_b).abs() > 1e-3 * float(dp_b.abs().max())).nonzero().reshape(-1)


def _ulp_of(pr, idx):
    import torch
    return torch.finfo(torch.float32).eps * pr[idx].abs().clamp_min(float(pr.pow(2).mean().sqrt()))


def net_grads(net, grads=None):
    return {nm: net.view(nm, net.grads if grads is None else grads).detach().cpu().double() for nm, _, _ in net.specs}


def single_shot_parity(run, oracle, compare, max_flips=8, loss_of=lambda net: net.loss_buf, pool_nets=lambda net: [net]):
    """Protocol of every whole-network GPU parity test.

    run() -> net: builds the network, one forward + backward on the device.
    oracle(net, pool_nudge) -> (ref, pool_inputs): the oracle step (forward + autograd backward) on the same inputs; ref[0] is
        the parameter dict whose `.grad`s are the oracle's gradients; pool_inputs = the tensors its max-poolings read,
        pool_nudge = None or what align_pool_ties returned (both flat lists over the pooled levels of pool_nets(net), in order).
    compare(net, ref): the test's assertions on everything but the gradients (prediction, loss, BatchNorm statistics).

    1. run() ONCE in deterministic mode (ops.set_deterministic: every cross-workgroup sum in a fixed order).  The oracle runs
       in float32 and, inside oracle.unet_ref.compute_dtype(float64), in float64; where its max-poolings and the device's
       disagree, the disagreement must be an identified rounding tie (align_pool_ties: candidates within 4 ulp, at most 8
       windows) and the oracle is re-run breaking those ties the way the device did.  compare() ONCE against the float32
       oracle; every gradient by the float64-anchored rule above (assert_grads_anchored).  No retry.
       Loss kinks (above) between device and oracle: identified from d(loss)/d(pred), each a rounding tie (<= KINK_ULP between the
       two predictions, at most 8 voxels), the oracle re-run on the device's side.
    2. run() once more on the default path (float atomics) and compare it WITH THE DETERMINISTIC RUN only: the arg-max masks of
       both device runs window by window (`_pool_choices`) -- differing windows must hold two candidates within 4 ulp of each
       other, at most `max_flips` of them --, the loss kinks the same way (predictions within KINK_ULP), the loss to 2e-6, and with
       identical choices every gradient within GRAD_K * o + GRAD_FLOOR of the deterministic one (accumulation-order noise is one
       more fp32 evaluation of the graph) and compare() against the oracle once more.  With identified flips the oracle is
       re-aligned to the ATOMICS run's own choices (float32 + float64) and compare() / assert_grads_anchored apply to that.
    Returns (net of the atomics run, number of windows flipped between the two device runs)."""
    import torch
    from synthsr_amd import ops
    from oracle import unet_ref as U

    def aligned_oracle(det_pool, det_kink, max_kink_ulp=KINK_ULP):
        rec = {'nudge': None, 'stop': None}

        def tap(out):   # oracle.unet_ref.prediction_tap: reads the oracle's prediction and its gradient, moves single voxels
            if rec['nudge'] is not None:
                out = out + rec['nudge'].to(out.dtype).view_as(out)
            out.retain_grad()
            rec['out'] = out
            if rec['stop'] is not None:   # voxels ON the kink in the device run (its derivative is exactly 0 there): no gradient
                out = torch.where(rec['stop'].view_as(out), out.detach(), out)
            return out

        def step(nudges):
            with U.prediction_tap(tap):
                ref_, pin = oracle(net, nudges)
            o = rec['out']
            g = o.grad.reshape(-1).float()
            if rec['stop'] is not None:
                g = torch.where(rec['stop'], torch.zeros_like(g), g)
            return ref_, pin, o.detach().reshape(-1).float(), g

        ref, pool_inputs, o_pr, o_dp = step(None)
        nudges, n_ties = align_pool_ties(det_pool, pool_inputs)
        if n_ties:
            print('single_shot_parity: %d max-pool rounding tie(s) between device and oracle, oracle re-run with the '
                  "device's choices" % n_ties)
            ref, pool_inputs, o_pr, o_dp = step(nudges)
            again = align_pool_ties(det_pool, [t if n is None else t + n.to(t.dtype) for t, n in zip(pool_inputs, nudges)])[1]
            assert again == 0, 'the nudged oracle still pools differently in %d windows' % again
        d_pr, d_dp = det_kink
        idx = _kink_disagreements(o_dp, d_dp)
        if idx.numel():
            assert d_pr is not None, 'the loss derivative differs grossly at %d voxels and the run kept no prediction' % idx.numel()
            ulp = _ulp_of(o_pr, idx)
            gap = d_pr[idx] - o_pr[idx]
            worst = float((gap.abs() / ulp).max())
            assert idx.numel() <= max_flips and worst <= max_kink_ulp, 'device and oracle sit on different sides of a kink of ' \
                'the loss at %d voxels whose predictions are up to %.1f ulp apart: not a rounding tie' % (idx.numel(), worst)
            print('single_shot_parity: %d loss-kink rounding tie(s) between device and oracle (predictions %.1f ulp apart), oracle '
                  "re-run on the device's side" % (idx.numel(), worst))
            side = torch.where(gap != 0, gap.sign(), (d_dp[idx] - o_dp[idx]).sign())
            rec['nudge'] = torch.zeros_like(o_pr)
            rec['nudge'][idx] = gap + 4.0 * ulp * side
            # a device prediction that hits its target EXACTLY has derivative 0 (sign(0)): no side of the kink to move the oracle
            # to -- the oracle's gradient is stopped at exactly those voxels instead (the planted-tie test produces them: its
            # targets sit one ulp from the deterministic run's predictions, where the atomics run may land)
            on_kink = d_dp[idx] == 0
            if bool(on_kink.any()):
                rec['stop'] = torch.zeros_like(o_pr, dtype=torch.bool)
                rec['stop'][idx[on_kink]] = True
                rec['nudge'][idx[on_kink]] = gap[on_kink]
            ref, pool_inputs, o_pr, o_dp = step(nudges)
            left = _kink_disagreements(o_dp, d_dp).numel()
            assert left == 0, 'the nudged oracle still disagrees with the device about %d loss kinks' % left
        return ref

    prev = ops.set_deterministic(True)
    try:
        net = run()
        assert ops.deterministic_status() == 1, 'an ordered wait timed out'
        det_pool = [c for n_ in pool_nets(net) for c in _pool_choices(n_)]
        det_kink = _kink_state(net)
        ref = aligned_oracle(det_pool, det_kink)
        with U.compute_dtype(torch.float64):
            ref64 = aligned_oracle(det_pool, det_kink)
        compare(net, ref)
        det_grads = net.grads.clone()
        det_loss = loss_of(net).clone()
        g32 = {nm: ref[0][nm].grad for nm, _, _ in net.specs}
        g64 = {nm: ref64[0][nm].grad for nm, _, _ in net.specs}
        assert_grads_anchored(net_grads(net), g32, g64, tag='deterministic run')
    finally:
        ops.set_deterministic(prev)
    net = run()
    flips = 0
    for l, ((m0, t0), (m1, t1)) in enumerate(zip(det_pool, [c for n_ in pool_nets(net) for c in _pool_choices(n_)])):
        diff = (_windows(m0) != _windows(m1)).any(1)
        n = int(diff.sum())
        if n:
            wall = _windows(t1)
            w = wall[diff]
            a = (w * _windows(m0)[diff]).sum(1)
            b = (w * _windows(m1)[diff]).sum(1)
            # an ulp of the larger of (candidate, rms of the tensor): see align_pool_ties
            rms = float(wall.float().pow(2).mean().sqrt())
            ulp = torch.finfo(torch.float32).eps * torch.maximum(a.abs(), b.abs()).clamp_min(rms)
            worst = float(((a - b).abs() / ulp).max())
            # What counts as a tie BETWEEN THE TWO DEVICE RUNS: two candidates closer than twice the distance the pooled tensor
            # itself moved from one run to the other cannot be ordered by either (each may have moved by that much).  The
            # first network's tensors move by a few ulp (float atomics in its BatchNorm statistics: the flat 4 ulp of round 5);
            # a DOWNSTREAM network (the frozen segmentation net reads the first one's prediction, which differs by up to ~36
            # ulp between runs, KINK_ULP above) inherits its input's noise -- round 6's final suite flipped one window of that
            # net whose candidates were 11.8 ulp apart.  Capped at 64 ulp: anything coarser is not rounding.
            moved = float((t1.float() - t0.float()).abs().max()) / (torch.finfo(torch.float32).eps * max(rms, 1e-30))
            tie = min(64.0, max(4.0, 2.0 * moved))
            assert worst <= tie, 'level %d: %d pooling windows changed their arg-max between the deterministic and the ' \
                'atomics run, candidates up to %.1f ulp apart while the tensor moved by %.1f ulp between the runs: not a ' \
                'rounding tie' % (l, n, worst, moved)
            flips += n
    assert flips <= max_flips, '%d pooling windows flipped (> %d)' % (flips, max_flips)
    assert abs(float(loss_of(net)) - float(det_loss)) <= 2e-6 * max(1.0, abs(float(det_loss)))
    a_pr, a_dp = _kink_state(net)            # loss kinks: voxels where the two device runs' loss derivatives differ grossly
    kidx = _kink_disagreements(a_dp, det_kink[1])
    if flips and kidx.numel():
        # a pooling window flipped between the runs: where it sits in a network DOWNSTREAM of the prediction (the frozen
        # segmentation net) it re-routes d(loss)/d(pred) over its whole receptive field -- hundreds of voxels differ grossly
        # between the two runs without any loss kink being involved, and the two effects cannot be told apart from the two
        # device runs alone.  The atomics run is then judged against its OWN oracle only (below): aligned to its pooling masks,
        # the oracle's d(loss)/d(pred) identifies its true kink ties (each <= KINK_ULP, at most max_flips) or fails.
        print('single_shot_parity: %d pooling flip(s) between the device runs; d(loss)/d(pred) differs at %d voxels: kinks are '
              'identified against the re-aligned oracle' % (flips, kidx.numel()))
    elif kidx.numel():
        assert a_pr is not None and det_kink[0] is not None, 'the loss derivative of the two device runs differs grossly at %d ' \
            'voxels and the runs kept no prediction' % kidx.numel()
        worst = float(((a_pr[kidx] - det_kink[0][kidx]).abs() / _ulp_of(det_kink[0], kidx)).max())
        assert kidx.numel() <= max_flips and worst <= KINK_ULP, '%d voxels changed the side of a loss kink between the deterministic ' \
            'and the atomics run, predictions up to %.1f ulp apart: not a rounding tie' % (kidx.numel(), worst)
        print('single_shot_parity: %d identified loss-kink flip(s) on the atomics path (predictions %.1f ulp apart)' % (kidx.numel(), worst))
        flips += int(kidx.numel())
    if flips == 0:
        compare(net, ref)                  # prediction / loss / BatchNorm statistics of the DEFAULT path against the oracle too
        assert_grads_anchored(net_grads(net, det_grads), g32, g64, tag='atomics run vs deterministic run', extra=net_grads(net))
    else:
        # the default path took the other side of `flips` identified rounding ties: its gradients legitimately differ from the
        # deterministic run's, so it gets its OWN oracle -- aligned with the atomics run's pooling masks and kink sides, float32
        # and float64 -- and the same anchored rule (this branch used to skip the gradient check)
        print('single_shot_parity: %d identified tie flip(s) (max-pool / loss kink) on the atomics path: oracle re-aligned to the '
              "atomics run's own choices" % flips)
        atom_pool = [c for n_ in pool_nets(net) for c in _pool_choices(n_)]
        atom_kink = (a_pr, a_dp)
        ref_a = aligned_oracle(atom_pool, atom_kink)
        with U.compute_dtype(torch.float64):
            ref64_a = aligned_oracle(atom_pool, atom_kink)
        compare(net, ref_a)
        assert_grads_anchored(net_grads(net), {nm: ref_a[0][nm].grad for nm, _, _ in net.specs},
                              {nm: ref64_a[0][nm].grad for nm, _, _ in net.specs}, tag='atomics run (own oracle alignment)')
    return net, flips
