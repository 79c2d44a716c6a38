"""GPU parity of the bf16 kernels (BASELINE.json configs[3] / [4]; csrc/conv_bf16.hip, bf16 pointwise kernels) through the
C ABI.  The reference is fp32 Keras, so parity is against the fp32 oracle (PyTorch-CPU) evaluated ON THE SAME
bf16-ROUNDED inputs / weights.  Stated bf16 tolerances: a conv output is one bf16 rounding (2^-9 relative) away from
the fp32-accumulated result -> 1e-2 of the tensor's range; fp32 weight gradients (no output rounding) 2e-3 of range."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def T():
    import torch
    assert torch.cuda.is_available()
    return torch


def close(a, b, rel, name=''):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(b.abs().max().item(), 1e-30)
    err = (a - b).abs().max().item() / scale
    assert err < rel, '%s: max rel err %.3e (scale %.3e)' % (name, err, scale)


def rbf(t):
    return t.bfloat16().float()


CASES = [((8, 8, 16), 24, 24), ((9, 7, 21), 24, 48), ((5, 6, 18), 48, 24), ((4, 4, 16), 72, 24), ((6, 5, 17), 8, 24),
         ((4, 8, 16), 96, 96), ((3, 4, 5), 192, 384), ((4, 4, 4), 576, 192), ((12, 12, 12), 32, 64), ((6, 6, 33), 144, 48),
         ((10, 10, 10), 384, 384), ((20, 20, 20), 96, 192), ((48, 40, 64), 24, 24), ((32, 48, 64), 72, 24),
         ((8, 8, 8), 256, 256), ((16, 16, 16), 64, 128)]


@pytest.mark.parametrize('shape,Cin,Cout', CASES)
def test_conv3d_bf16_fwd_dgrad_wgrad(T, shape, Cin, Cout):
    torch = T
    from synthsr_amd import ops
    from oracle import unet_ref as U
    g = torch.Generator().manual_seed(Cin * 1000 + Cout)
    x = rbf(torch.randn(*shape, Cin, generator=g))
    w = rbf(torch.randn(3, 3, 3, Cin, Cout, generator=g) / np.sqrt(27 * Cin))
    b = torch.randn(Cout, generator=g)
    dy = rbf(torch.randn(*shape, Cout, generator=g))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = U.conv3d_same(xr, wr, b)
    yr_elu = torch.nn.functional.elu(yr)
    yr.backward(dy)
    xd, wd, bd, dyd = x.cuda().bfloat16(), w.cuda(), b.cuda(), dy.cuda().bfloat16()
    wp = ops.pack_conv_weights_bf16(wd, 0)
    close(ops.conv3d_bf16(xd, wp, bd, Cout, act=0).float(), yr, 1e-2, 'fwd linear')
    stats = torch.zeros(2 * Cout, device='cuda')
    y = ops.conv3d_bf16(xd, wp, bd, Cout, act=1, stats=stats)
    close(y.float(), yr_elu, 1e-2, 'fwd elu')
    yf = y.float().reshape(-1, Cout)
    close(stats[:Cout], yf.mean(0), 2e-4, 'batch mean (of the rounded output)')
    close(stats[Cout:], yf.var(0, unbiased=False), 2e-4, 'batch variance')
    if Cout % 8 == 0:
        wpd = ops.pack_conv_weights_bf16(wd, 1)
        close(ops.conv3d_bf16(dyd, wpd, None, Cin, act=0).float(), xr.grad, 1e-2, 'dgrad')
        below = rbf(torch.nn.functional.elu(torch.randn(*shape, Cin, generator=g)))
        deriv = torch.where(below > 0, torch.ones_like(below), below + 1)
        close(ops.conv3d_bf16(dyd, wpd, None, Cin, act=2, below=below.cuda().bfloat16()).float(), xr.grad * deriv, 1e-2,
              "dgrad * elu'")
        dw, db = torch.zeros(3, 3, 3, Cin, Cout, device='cuda'), torch.zeros(Cout, device='cuda')
        ops.conv3d_wgrad_bf16(xd, dyd, dw, db)
        close(dw, wr.grad, 2e-3, 'wgrad')
        close(db, dy.reshape(-1, Cout).sum(0), 2e-3, 'dbias')


def test_conv3d_bf16_transpose_detecting(T):
    """asymmetric one-hot kernels: a swapped tap / channel / fragment-lane mapping cannot hide"""
    torch = T
    from synthsr_amd import ops
    x = rbf(torch.randn(6, 7, 19, 24))
    for tap, ci, co in [((0, 1, 2), 3, 17), ((2, 0, 1), 23, 0), ((1, 1, 1), 5, 5), ((2, 2, 0), 8, 23)]:
        w = torch.zeros(3, 3, 3, 24, 24)
        w[tap[0], tap[1], tap[2], ci, co] = 1.0
        y = ops.conv3d_bf16(x.cuda().bfloat16(), ops.pack_conv_weights_bf16(w.cuda(), 0), None, 24, act=0).float().cpu()
        xp = torch.nn.functional.pad(x[..., ci], (1, 1, 1, 1, 1, 1))
        exp = xp[tap[0]:tap[0] + 6, tap[1]:tap[1] + 7, tap[2]:tap[2] + 19]
        assert torch.equal(y[..., co], exp), (tap, ci, co)
        assert int((y != 0).sum()) == int((exp != 0).sum())
    # weight gradient of one-hot dz / x: exactly one (tap, ci, co) entry per shifted overlap
    xs = torch.zeros(6, 7, 19, 24)
    dz = torch.zeros(6, 7, 19, 24)
    xs[2, 3, 5, 7] = 1.0
    dz[3, 3, 4, 13] = 2.0        # x sits at dz position + (-1, 0, +1) -> tap (0, 1, 2)
    dw = torch.zeros(3, 3, 3, 24, 24, device='cuda')
    ops.conv3d_wgrad_bf16(xs.cuda().bfloat16(), dz.cuda().bfloat16(), dw)
    exp = torch.zeros(3, 3, 3, 24, 24)
    exp[0, 1, 2, 7, 13] = 2.0
    assert torch.equal(dw.cpu(), exp)


def test_f32_to_bf16_pad(T):
    torch = T
    from synthsr_amd import ops
    x = torch.randn(5, 6, 7, 2)
    y = ops.to_bf16_pad(x.cuda(), 8).float().cpu()
    assert torch.equal(y[..., :2], rbf(x)) and float(y[..., 2:].abs().max()) == 0.0
