"""fp32 convolutions through three bf16 pieces per operand (csrc/conv_split.hip, ops.set_conv_arithmetic('split'), the default)
are fp32 computations: measured against a FLOAT64 convolution of the same fp32 inputs they are as accurate as the fp32 matrix
instructions (csrc/conv3d.hip), on benign data and on data chosen to expose a bf16 short-cut (large common offsets, a wide
dynamic range, exact cancellation).  The reference computes these layers in fp32 on TensorFlow (ext/neuron/models.py:256-498);
float64 is the common yardstick, the tolerance of every assertion is written next to it."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref64(x, w, b=None):
    xi = x.double().cpu().permute(3, 0, 1, 2)[None]
    wi = w.double().cpu().permute(4, 3, 0, 1, 2)
    return F.conv3d(xi, wi, None if b is None else b.double().cpu(), padding=1)[0].permute(1, 2, 3, 0)


def _wgrad64(x, dy):
    xi = x.double().cpu().permute(3, 0, 1, 2)[None]
    g = dy.double().cpu().permute(3, 0, 1, 2)[None]
    w = torch.zeros(dy.shape[3], x.shape[3], 3, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv3d(xi, w, None, padding=1).backward(g)
    return w.grad.permute(2, 3, 4, 1, 0)


def _err(y, ref):
    d = y.double().cpu() - ref
    scale = float(ref.pow(2).mean().sqrt())
    return float(d.abs().max()) / scale, float(d.pow(2).mean().sqrt()) / scale


def _plan(shape, cin, cout):
    import ctypes
    from synthsr_amd import _lib
    out = (ctypes.c_int64 * 8)()
    from synthsr_amd import ops
    _lib.check(_lib.load().synthsr_conv3d_plan(ops.conv_ctx_host(), _lib.i3(shape), cin, cout, 1, out), 'plan')
    return [int(v) for v in out]


def _is_split(shape, cin, cout):
    return _plan(shape, cin, cout)[2] <= -100


def _data(kind, D, ci, co, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    S = (D, D, D) if isinstance(D, int) else tuple(D)
    x = torch.randn(*S, ci, generator=g)
    dy = torch.randn(*S, co, generator=g)
    w = torch.randn(3, 3, 3, ci, co, generator=g) * 0.05
    if kind == 'offset':      # a large common offset: bf16 alone would lose the signal (8 significand bits of 100 + noise)
        x = x * 0.01 + 100.0
        dy = dy * 0.01 - 37.0
    elif kind == 'range':     # seven decades of magnitude between channels
        x = x * torch.logspace(-3, 4, ci)
        dy = dy * torch.logspace(-2, 3, co)
        w = w * torch.logspace(-2, 2, ci)[:, None]
    elif kind == 'cancel':    # weights that sum to ~0 over the taps on a nearly constant input: the result is all rounding
        x = 1.0 + 1e-4 * x
        w = w - w.mean((0, 1, 2), keepdim=True)
    return x.cuda(), dy.cuda(), w.cuda(), (torch.randn(co, generator=g) * 0.1).cuda()


@pytest.mark.parametrize('kind', ['normal', 'offset', 'range', 'cancel'])
@pytest.mark.parametrize('D,ci,co', [(48, 24, 24), (48, 24, 48), (40, 48, 48), (40, 96, 48), (40, 96, 96), (48, 16, 16),
                                     ((42, 38, 50), 24, 24), ((38, 42, 50), 48, 48), (20, 192, 192), ((18, 21, 23), 96, 192)])
def test_split_conv_is_as_accurate_as_the_fp32_mfma_kernels(D, ci, co, kind):
    """(cubes, a one-column-tile layer, volumes whose sizes are no multiples of the 4x4x16 tile: masked loads / stores, and --
    the last two -- layers that run as split-K halves, csrc/conv_split.hip conv3d_split_fwd2_kernel<..., KS = 2>)"""
    from synthsr_amd import ops
    shape = (D, D, D) if isinstance(D, int) else tuple(D)
    x, dy, w, b = _data(kind, D, ci, co, seed=sum(shape) + ci + co)
    refs = (_ref64(x, w, b), _ref64(dy, torch.flip(w, (0, 1, 2)).transpose(3, 4)), _wgrad64(x, dy), dy.double().cpu().sum((0, 1, 2)))
    res = {}
    prev = ops.conv_arithmetic()
    # deterministic mode: the weight / bias gradients are sums of per-workgroup partials; with atomics their last bits (and so
    # the ratio of two tiny errors) would change from run to run -- ordered sums make this test reproducible
    prev_det = ops.set_deterministic(True)
    try:
        for mode in ('fp32_mfma', 'split'):
            ops.set_conv_arithmetic(mode)
            assert _is_split(shape, ci, co) == (mode == 'split')
            if mode == 'split' and max(shape) <= 24:
                assert _plan(shape, ci, co)[5] == 2   # the forward launch runs as split-K halves (20^3 192 -> 192: the data gradient too)
            wp, wpd = ops.pack_conv_weights(w, shape, 0), ops.pack_conv_weights(w, shape, 1)
            y = ops.conv3d(x, wp, b, co, 0)
            dx = ops.conv3d(dy, wpd, None, ci, 0)
            dw, db = torch.zeros_like(w), torch.zeros_like(b)
            ops.conv3d_wgrad(x, dy, dw, db)
            res[mode] = [_err(t, r) for t, r in zip((y, dx, dw, db), refs)]
        assert ops.deterministic_status() == 1
    finally:
        ops.set_deterministic(prev_det)
        ops.set_conv_arithmetic(prev)
    for k, name in enumerate(('forward', 'data gradient', 'weight gradient', 'bias gradient')):
        (nmax, nrms), (smax, srms) = res['fp32_mfma'][k], res['split'][k]
        # (1) not less accurate than the fp32 matrix instructions: rms error within 1.5x (+ 1e-8 for the cases where both are
        #     ~0), worst element within 2.5x (the two arithmetics group the partial sums of a weight / bias gradient over 10^5
        #     voxels differently: measured ratios lie between 0.8 and 1.3);  (2) an fp32 result in absolute terms: rms error
        #     below 4e-6 of the result's rms (bf16 inputs alone would be at 4e-3)
        assert srms <= 1.5 * nrms + 1e-8, (name, kind, res)
        assert smax <= 2.5 * nmax + 1e-7, (name, kind, res)
        assert srms < 4e-6 or kind in ('cancel', 'offset'), (name, kind, res)


@pytest.mark.parametrize('D,cs,cl,co', [(48, 24, 48, 24), (40, 48, 96, 48)])
def test_split_folded_decoder_conv_vs_float64(D, cs, cl, co):
    """the up-sampled channel range of a folded decoder conv (8 parity 2x2x2 convs on the low-resolution tensor): forward and data
    gradient in split arithmetic against a float64 evaluation of conv3(UpSampling3D(2)(lo)) and its gradient, next to the fp32
    MFMA kernels (the weight gradient of this part: test_split_folded_weight_gradient_vs_float64)"""
    from synthsr_amd import ops
    g = torch.Generator(device='cpu').manual_seed(D + cl)
    lo = torch.randn(D, D, D, cl, generator=g).cuda()
    dz = torch.randn(2 * D, 2 * D, 2 * D, co, generator=g).cuda()
    w = (torch.randn(3, 3, 3, cs + cl, co, generator=g) * 0.05).cuda()
    up = lo.double().cpu().repeat_interleave(2, 0).repeat_interleave(2, 1).repeat_interleave(2, 2).requires_grad_(True)
    wu = w[:, :, :, cs:].double().cpu()
    y = F.conv3d(up.permute(3, 0, 1, 2)[None], wu.permute(4, 3, 0, 1, 2), None, padding=1)[0].permute(1, 2, 3, 0)
    y.backward(dz.double().cpu())
    dlo_ref = up.grad.reshape(D, 2, D, 2, D, 2, cl).sum((1, 3, 5))
    res = {}
    prev = ops.conv_arithmetic()
    try:
        for mode in ('fp32_mfma', 'split'):
            ops.set_conv_arithmetic(mode)
            wp_u = ops.pack_conv_weights_ex(w, (D, D, D), cs, cl, 0, up=True)
            wpd_u = ops.pack_conv_weights_ex(w, (D, D, D), cs, cl, 1, up=True)
            res[mode] = (_err(ops.conv3d_up(lo, wp_u, None, None, co, 0), y.detach()), _err(ops.conv3d_up_dgrad(dz, wpd_u, cl), dlo_ref))
    finally:
        ops.set_conv_arithmetic(prev)
    for k in range(2):
        (nmax, nrms), (smax, srms) = res['fp32_mfma'][k], res['split'][k]
        assert srms <= 1.25 * nrms and smax <= 2.0 * nmax and srms < 1.5e-6, res   # deterministic kernels: tight bounds


def test_split_pieces_reconstruct_fp32_exactly():
    """a = a0 + a1 + a2 with every piece a bf16 number: checked through the conv itself -- a 1-tap identity kernel must return
    its input to the last bit (x * 1.0 is exact in every partial product, the three pieces add up in the fp32 accumulator)"""
    from synthsr_amd import ops
    D, C = 48, 24
    g = torch.Generator(device='cpu').manual_seed(3)
    x = (torch.randn(D, D, D, C, generator=g) * torch.logspace(-6, 6, C)).cuda()
    w = torch.zeros(3, 3, 3, C, C, device='cuda')
    w[1, 1, 1] = torch.eye(C)
    assert ops.conv_arithmetic() == 'split' and _is_split((D, D, D), C, C)
    y = ops.conv3d(x, ops.pack_conv_weights(w, (D, D, D), 0), None, C, 0)
    assert torch.equal(y, x)


def test_split_one_hot_kernels_shift_exactly():
    """asymmetric one-hot kernels at a split-eligible size: forward, data gradient and weight gradient must be exact shifts /
    correlations of the input (catches a swapped tap, channel or lane-voxel mapping that random data could average out; the
    weight 1.0 and the three pieces of every activation multiply exactly, so the forward results are bit-exact)"""
    from synthsr_amd import ops
    D, C = 48, 24
    g = torch.Generator(device='cpu').manual_seed(5)
    x = torch.randn(D, D, D, C, generator=g)
    assert ops.conv_arithmetic() == 'split' and _is_split((D, D, D), C, C)
    for tap, ci, co in [((0, 1, 2), 3, 17), ((2, 0, 1), 23, 0), ((1, 1, 1), 5, 5), ((2, 2, 0), 8, 16)]:
        w = torch.zeros(3, 3, 3, C, C)
        w[tap[0], tap[1], tap[2], ci, co] = 1.0
        wd = w.cuda()
        y = ops.conv3d(x.cuda(), ops.pack_conv_weights(wd, (D, D, D), 0), None, C, 0).cpu()
        # y[v, co] = x[v + tap - 1, ci] (zero outside the volume)
        xp = torch.nn.functional.pad(x[..., ci], (1, 1, 1, 1, 1, 1))
        want = xp[tap[0]:tap[0] + D, tap[1]:tap[1] + D, tap[2]:tap[2] + D]
        assert torch.equal(y[..., co], want), (tap, ci, co)
        others = [c for c in range(C) if c != co]
        assert float(y[..., others].abs().max()) == 0.0   # nothing leaks into another output channel
        # data gradient: dx[u, ci] = dy[u - (tap - 1), co]
        dy = torch.randn(D, D, D, C, generator=g)
        dx = ops.conv3d(dy.cuda(), ops.pack_conv_weights(wd, (D, D, D), 1), None, C, 0).cpu()
        dyp = torch.nn.functional.pad(dy[..., co], (1, 1, 1, 1, 1, 1))
        wantd = dyp[2 - tap[0]:2 - tap[0] + D, 2 - tap[1]:2 - tap[1] + D, 2 - tap[2]:2 - tap[2] + D]
        assert torch.equal(dx[..., ci], wantd), (tap, ci, co)
        # weight gradient at that tap / channel pair = <shifted x, dy> (a 110 592-term fp32 sum: 2e-5 of its scale)
        dw = torch.zeros(3, 3, 3, C, C, device='cuda')
        ops.conv3d_wgrad(x.cuda(), dy.cuda(), dw)
        ref = float((want.double() * dy[..., co].double()).sum())
        scale = float(want.double().pow(2).sum().sqrt() * dy[..., co].double().pow(2).sum().sqrt())
        assert abs(float(dw[tap[0], tap[1], tap[2], ci, co]) - ref) < 2e-5 * scale, (tap, ci, co)


def test_split9_all_nine_products():
    """ops.set_conv_arithmetic('split9'): the same kernels with all nine partial products -- every fp32 product exact; against
    float64 at least as accurate as the fp32 matrix instructions, forward / data gradient / weight gradient, plain and folded"""
    from synthsr_amd import ops
    D, ci, co = 48, 24, 48
    shape = (D, D, D)
    x, dy, w, b = _data('normal', D, ci, co, seed=77)
    refs = (_ref64(x, w, b), _ref64(dy, torch.flip(w, (0, 1, 2)).transpose(3, 4)), _wgrad64(x, dy))
    res = {}
    prev = ops.conv_arithmetic()
    prev_det = ops.set_deterministic(True)
    try:
        for mode in ('fp32_mfma', 'split9'):
            ops.set_conv_arithmetic(mode)
            assert ops.conv_arithmetic() == mode and _is_split(shape, ci, co) == (mode == 'split9')
            wp, wpd = ops.pack_conv_weights(w, shape, 0), ops.pack_conv_weights(w, shape, 1)
            dw = torch.zeros_like(w)
            ops.conv3d_wgrad(x, dy, dw)
            res[mode] = [_err(t, r) for t, r in zip((ops.conv3d(x, wp, b, co, 0), ops.conv3d(dy, wpd, None, ci, 0), dw), refs)]
    finally:
        ops.set_deterministic(prev_det)
        ops.set_conv_arithmetic(prev)
    for k in range(3):
        (nmax, nrms), (smax, srms) = res['fp32_mfma'][k], res['split9'][k]
        assert srms <= 1.25 * nrms and smax <= 2.0 * nmax and srms < 1.5e-6, res


@pytest.mark.parametrize('act', [0, 1])
def test_split_forward_epilogues_and_fused_statistics(act):
    """bias + ELU, the addend / ELU'-gating epilogues and the BatchNorm statistics of the output, against torch on the device"""
    from synthsr_amd import ops
    D, ci, co = 48, 24, 24
    x, dy, w, b = _data('normal', D, ci, co, seed=11)
    shape = (D, D, D)
    wp = ops.pack_conv_weights(w, shape, 0)
    lin = _ref64(x, w, b)
    want = (F.elu(lin) if act else lin)
    def ok(t, ref):   # fp32 rounding of a 648-term sum: worst element 2e-5, rms 1.5e-6 of the result's rms
        mx, rms = _err(t, ref)
        return mx < 2e-5 and rms < 1.5e-6

    y = ops.conv3d(x, wp, b, co, act)
    assert ok(y, want)
    add = torch.randn(D, D, D, co, device='cuda')
    ya = ops.conv3d_add(x, wp, b, add, co, act)
    wa = lin + add.double().cpu()
    assert ok(ya, F.elu(wa) if act else wa)
    below = F.elu(torch.randn(D, D, D, co, device='cuda'))
    yg = ops.conv3d_add(x, wp, None, below, co, 2)   # conv * ELU'(below), ELU' through its output
    wg = _ref64(x, w) * torch.where(below > 0, torch.ones_like(below), below + 1).double().cpu()
    assert ok(yg, wg)
    stats, ws = torch.zeros(2 * co, device='cuda'), torch.zeros(2 * co, dtype=torch.float64, device='cuda')
    ys = ops.conv3d_stats(x, wp, b, co, stats, ws, act)
    assert torch.equal(ys, y)
    flat = want.reshape(-1, co)
    assert float((stats[:co].double().cpu() - flat.mean(0)).abs().max()) < 2e-6
    assert float((stats[co:].double().cpu() - flat.var(0, unbiased=False)).abs().max()) < 2e-6 * float(flat.var(0).max())


def test_network_step_agrees_between_the_two_arithmetics():
    """one training step of a small U-Net under both arithmetics: same loss and gradients to fp32 accuracy (and the packed
    weights follow the mode switch)"""
    from synthsr_amd import ops
    from synthsr_amd.unet import unet
    shape = (64, 64, 64)
    res = {}
    prev = ops.conv_arithmetic()
    try:
        for mode in ('fp32_mfma', 'split'):
            ops.set_conv_arithmetic(mode)
            net = unet(nb_features=24, input_shape=list(shape) + [2], nb_levels=3, conv_size=3, nb_labels=1, feat_mult=2,
                       nb_conv_per_level=2, batch_norm=-1, activation='elu', final_pred_activation='linear', seed=5)
            g = torch.Generator(device='cpu').manual_seed(9)
            x = torch.randn(*shape, 2, generator=g).cuda()
            target = torch.randn(*shape, 1, generator=g).cuda()
            prev_det = ops.set_deterministic(True)
            try:
                loss = net.loss_l1(x, target.reshape(-1))[0].clone()
                net.backward()
            finally:
                ops.set_deterministic(prev_det)
            res[mode] = (float(loss.item()), net.grads.clone())
    finally:
        ops.set_conv_arithmetic(prev)
    (l0, g0), (l1, g1) = res['fp32_mfma'], res['split']
    assert abs(l0 - l1) < 2e-6 * abs(l0)
    cos = float((g0 * g1).sum() / (g0.norm() * g1.norm()))
    assert cos > 1 - 1e-6, cos
    assert float((g0 - g1).abs().max()) < 2e-3 * float(g0.abs().max())   # max-pool ties may route a few gradients differently


def _up_wgrad64(lo, dz):
    """float64 weight gradient of conv3(UpSampling3D(2)(lo)) w.r.t. its [3,3,3,Cl,Cout] kernel"""
    up = lo.double().cpu().repeat_interleave(2, 0).repeat_interleave(2, 1).repeat_interleave(2, 2)
    return _wgrad64(up, dz)


@pytest.mark.parametrize('lo_shape,cl,co', [
    ((16, 16, 32), 48, 24), ((8, 8, 16), 96, 48), ((4, 4, 16), 16, 24), ((10, 10, 10), 32, 24), ((5, 7, 9), 16, 48),
    ((18, 14, 34), 48, 24), ((20, 20, 20), 192, 96), ((12, 10, 18), 64, 72)])
def test_split_folded_weight_gradient_vs_float64(lo_shape, cl, co):
    """(round 6) the weight gradient of the up-sampled channel range of a folded decoder conv in split arithmetic
    (csrc/conv_split.hip conv3d_split_upwgrad_kernel: the eight waves of a workgroup = the eight output parities over one staged
    low-resolution halo; models.py:426-444) against a float64 evaluation of the weight gradient of conv3(UpSampling3D(2)(lo)),
    next to the fp32 matrix instructions: few tiles, ragged volumes, several column chunks, TWO volumes accumulated into one dW as
    a batch does; ordered sums and float atomics."""
    from synthsr_amd import ops
    g = torch.Generator(device='cpu').manual_seed(sum(lo_shape) + 7 * cl + co)
    hi = tuple(2 * v for v in lo_shape)
    los = [torch.randn(*lo_shape, cl, generator=g) for _ in range(2)]
    dzs = [torch.randn(*hi, co, generator=g) * s for s in (1.0, 1.7)]
    cs = 8
    ref = sum(_up_wgrad64(lo, dz) for lo, dz in zip(los, dzs))
    los, dzs = [v.cuda() for v in los], [v.cuda() for v in dzs]
    res = {}
    prev = ops.conv_arithmetic()
    try:
        for mode in ('fp32_mfma', 'split'):
            ops.set_conv_arithmetic(mode)
            assert ops.conv_runs_split('conv3d_up_wgrad', lo_shape, cl, co) == (mode == 'split')
            for det in (True, False):
                prev_det = ops.set_deterministic(det)
                try:
                    dw = torch.zeros(3, 3, 3, cs + cl, co, device='cuda')
                    dwc = torch.empty(8, 27, cl, co, device='cuda')
                    for lo, dz in zip(los, dzs):
                        ops.conv3d_up_wgrad(lo, dz, dwc, dw, cs)
                    assert not det or ops.deterministic_status() == 1
                finally:
                    ops.set_deterministic(prev_det)
                assert float(dw[:, :, :, :cs].abs().max()) == 0.0          # the skip channels' rows are not touched
                res[mode, det] = _err(dw[:, :, :, cs:], ref)
    finally:
        ops.set_conv_arithmetic(prev)
    for det in (True, False):
        (nmax, nrms), (smax, srms) = res['fp32_mfma', det], res['split', det]
        # the bounds of test_split_weight_gradient_small_and_ragged_layers_vs_float64 (sums of <= 2 x 8 x 9 216 products)
        assert srms < 2e-6 and smax < 2e-5, (det, res)
        assert srms <= 1.5 * nrms + 2e-8 and smax <= 2.5 * nmax + 2e-7, (det, res)


@pytest.mark.parametrize('shape,ci,co', [
    ((16, 16, 32), 24, 24), ((32, 16, 16), 24, 24), ((8, 8, 16), 24, 24), ((4, 4, 16), 24, 24), ((20, 20, 20), 24, 24),
    ((10, 10, 10), 24, 24), ((18, 14, 34), 24, 24), ((5, 7, 9), 24, 24),          # Cin = Cout = 24: all 24 input channels per workgroup
    ((16, 16, 32), 48, 48), ((8, 8, 16), 96, 48), ((12, 10, 18), 48, 96), ((10, 10, 10), 192, 192),   # Cin % 16 == 0, Cout % 48 == 0
    ((16, 16, 32), 24, 48), ((8, 12, 16), 72, 24), ((6, 6, 6), 8, 24)])            # 8 input channels per workgroup
def test_split_weight_gradient_small_and_ragged_layers_vs_float64(shape, ci, co):
    """every split weight-gradient kernel at the sizes the small whole-network tests run it (since round 4 the split kernel takes
    layers of ANY size: include/synthsr_hip_tuning.h option 11) -- few tiles, volumes that are no multiples of the 4x4x16 tile, and
    TWO volumes accumulated into one dW / dbias as a batch does (synthsr_amd/unet.py: set_batch) -- against a float64 evaluation,
    next to the fp32 matrix instructions; ordered sums and float atomics."""
    from synthsr_amd import ops
    g = torch.Generator(device='cpu').manual_seed(sum(shape) + 7 * ci + co)
    xs = [torch.randn(*shape, ci, generator=g) for _ in range(2)]
    dys = [torch.randn(*shape, co, generator=g) * s for s in (1.0, 1.7)]
    ref_w = sum(_wgrad64(x, dy) for x, dy in zip(xs, dys))
    ref_b = sum(dy.double().sum((0, 1, 2)) for dy in dys)
    xs, dys = [x.cuda() for x in xs], [dy.cuda() for dy in dys]
    res = {}
    prev = ops.conv_arithmetic()
    try:
        for mode in ('fp32_mfma', 'split'):
            ops.set_conv_arithmetic(mode)
            assert ops.conv_runs_split('conv3d_wgrad', shape, ci, co) == (mode == 'split')
            for det in (True, False):
                prev_det = ops.set_deterministic(det)
                try:
                    dw, db = torch.zeros(3, 3, 3, ci, co, device='cuda'), torch.zeros(co, device='cuda')
                    for x, dy in zip(xs, dys):
                        ops.conv3d_wgrad(x, dy, dw, db)
                    assert not det or ops.deterministic_status() == 1
                finally:
                    ops.set_deterministic(prev_det)
                res[mode, det] = (_err(dw, ref_w), _err(db, ref_b))
    finally:
        ops.set_conv_arithmetic(prev)
    for det in (True, False):
        for k, name in enumerate(('weight gradient', 'bias gradient')):
            (nmax, nrms), (smax, srms) = res['fp32_mfma', det][k], res['split', det][k]
            # an fp32 result in absolute terms (sums of <= 2 x 16 384 products: rms error below 2e-6 of the result's rms, worst
            # element 2e-5), and not less accurate than the fp32 matrix instructions (rms 1.5x, worst element 2.5x; + floors)
            assert srms < 2e-6 and smax < 2e-5, (name, det, res)
            assert srms <= 1.5 * nrms + 2e-8 and smax <= 2.5 * nmax + 2e-7, (name, det, res)
