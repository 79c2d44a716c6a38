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
        x = net.saved['enc'][l][-1]
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


def single_shot_parity(run, check, max_flips=8, atomics_tol=1e-3, loss_of=lambda net: net.loss_buf,
                       pool_nets=lambda net: [net]):
    """Protocol of every whole-network GPU parity test (replaces the former retry-on-failure decorator).

    1. `run()` (builds the network, one forward + backward, returns the net) ONCE in deterministic mode
       (ops.set_deterministic: every cross-workgroup sum in a fixed order) and `check(net)` -- the comparison with the
       oracle at the test's tolerances -- ONCE.  No retry: a failure here is a failure.
    2. `run()` once more on the default path (float atomics).  The network's max-pooling is discontinuous, so the
       accumulation-order noise of the BatchNorm statistics can flip an arg-max between two values within float32 rounding
       of each other, which moves that level's gradients by a few 1e-3 of their range.  That is the ONLY difference this
       function tolerates, and it has to be IDENTIFIED: the arg-max masks of both runs are read back from the device
       (`_pool_choices`); with identical masks the atomics run must pass `check` and agree with the deterministic gradients
       to `atomics_tol` of each tensor's range; with differing masks every differing window must hold two candidates
       within 4 ulp of each other and there may be at most `max_flips` of them -- anything else fails.
    Returns (net of the atomics run, number of flipped windows)."""
    import torch
    from synthsr_amd import ops
    prev = ops.set_deterministic(True)
    try:
        net = run()
        check(net)
        assert ops.deterministic_status() == 1, 'an ordered wait timed out'
        det_grads = net.grads.clone()
        det_loss = loss_of(net).clone()
        det_pool = [c for n_ in pool_nets(net) for c in _pool_choices(n_)]
    finally:
        ops.set_deterministic(prev)
    net = run()
    flips = 0
    for l, ((m0, _), (m1, t1)) in enumerate(zip(det_pool, [c for n_ in pool_nets(net) for c in _pool_choices(n_)])):
        diff = (_windows(m0) != _windows(m1)).any(1)
        n = int(diff.sum())
        if n:
            w = _windows(t1)[diff]
            a = (w * _windows(m0)[diff]).sum(1)
            b = (w * _windows(m1)[diff]).sum(1)
            ulp = torch.finfo(torch.float32).eps * torch.maximum(a.abs(), b.abs()).clamp_min(1e-30)
            worst = float(((a - b).abs() / ulp).max())
            assert worst <= 4.0, 'level %d: %d pooling windows changed their arg-max between the deterministic and the ' \
                'atomics run, candidates up to %.1f ulp apart: not a rounding tie' % (l, n, worst)
            flips += n
    assert flips <= max_flips, '%d pooling windows flipped (> %d)' % (flips, max_flips)
    assert abs(float(loss_of(net)) - float(det_loss)) <= 2e-6 * max(1.0, abs(float(det_loss)))
    if flips == 0:
        check(net)
        for nm, _, kind in net.specs:  # kernels: atomics_tol of the tensor's range; sums of cancelling terms (biases, BN): 4x
            a, b = net.view(nm, net.grads), net.view(nm, det_grads)
            err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-3 * float(det_grads.abs().max()) + 1e-30))
            assert err < atomics_tol * (1 if kind in ('kernel', 'head_w') else 4), 'atomics vs deterministic gradient of ' \
                '%s: %.2e of its range with identical pooling choices' % (nm, err)
    else:
        print('single_shot_parity: %d identified max-pool tie flip(s) on the atomics path' % flips)
    return net, flips
