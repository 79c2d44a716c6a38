#!/usr/bin/env python
"""fp32 convs through three bf16 pieces per operand (csrc/conv_split.hip) against the fp32-MFMA kernels (csrc/conv3d.hip):
error of both versus a float64 convolution, and time per layer shape.

    python tools/split_check.py [--acc] [--time] [--reps 10]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from synthsr_amd import _lib, ops  # noqa: E402

def set_split(on):
    ops.set_conv_arithmetic('split' if on else 'fp32_mfma')


def plan_is_split(shape, cin, cout):
    out = (ctypes.c_int64 * 8)()
    _lib.check(_lib.load().synthsr_conv3d_plan(ops.conv_ctx_host(), _lib.i3(shape), cin, cout, 1, out), 'plan')
    return int(out[2]) <= -100


def ref64(x, w, b=None):
    """float64 'same' conv of x [D,D,D,Cin] with the Keras kernel w [3,3,3,Cin,Cout] on the host"""
    xi = x.double().cpu().permute(3, 0, 1, 2)[None]
    wi = w.double().cpu().permute(4, 3, 0, 1, 2)
    y = F.conv3d(xi, wi, None if b is None else b.double().cpu(), padding=1)
    return y[0].permute(1, 2, 3, 0)


def wgrad64(x, dy):
    xi = x.double().cpu().permute(3, 0, 1, 2)[None]
    g = dy.double().cpu().permute(3, 0, 1, 2)[None]
    w = torch.zeros(dy.shape[3], x.shape[3], 3, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv3d(xi, w, None, padding=1).backward(g)
    return w.grad.permute(2, 3, 4, 1, 0)


def errs(y, ref):
    d = (y.double().cpu() - ref)
    scale = ref.pow(2).mean().sqrt().item()
    return d.abs().max().item() / scale, d.pow(2).mean().sqrt().item() / scale


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def accuracy():
    torch.manual_seed(0)
    print('errors relative to the rms of the float64 result: max / rms')
    print('%-22s %-24s %-24s' % ('layer', 'fp32 MFMA', 'split bf16 x6'))
    for D, ci, co in [(32, 24, 24), (32, 48, 48), (48, 24, 48), (40, 96, 96), (32, 8, 24)]:
        shape = (D, D, D)
        x = torch.randn(D, D, D, ci, device='cuda') * torch.rand(D, D, D, 1, device='cuda') * 2
        w = torch.randn(3, 3, 3, ci, co, device='cuda') * 0.05
        b = torch.randn(co, device='cuda')
        dy = torch.randn(D, D, D, co, device='cuda')
        r = ref64(x, w, b)
        rd = ref64(dy, torch.flip(w, (0, 1, 2)).transpose(3, 4))
        rw = wgrad64(x, dy)
        res = {}
        for mode in (0, 1):
            set_split(mode)
            wp, wpd = ops.pack_conv_weights(w, shape, 0), ops.pack_conv_weights(w, shape, 1)
            y = ops.conv3d(x, wp, b, co, 0)
            dx = ops.conv3d(dy, wpd, None, ci, 0)
            dw = torch.zeros_like(w)
            ops.conv3d_wgrad(x, dy, dw)
            res[mode] = (errs(y, r), errs(dx, rd), errs(dw, rw), plan_is_split(shape, ci, co))
        set_split(0)
        for k, nm in enumerate(('fwd', 'dgrad', 'wgrad')):
            print('%3d^3 %3d->%-3d %-6s   %.3e / %.3e    %.3e / %.3e   %s' % (
                D, ci, co, nm, res[0][k][0], res[0][k][1], res[1][k][0], res[1][k][1], '' if res[1][3] else '(not split)'))


def timing(reps, only=''):
    print('%-18s %12s %12s %12s   (ms: fp32 MFMA -> split)' % ('layer', 'fwd', 'dgrad', 'wgrad'))
    for D, ci, co in [(160, 24, 24), (80, 24, 48), (80, 48, 48), (80, 48, 24), (40, 48, 96), (40, 96, 96), (40, 96, 48)]:
        if only and '%d_%d_%d' % (D, ci, co) not in only.split(','):
            continue
        shape = (D, D, D)
        x = torch.randn(D, D, D, ci, device='cuda')
        w = torch.randn(3, 3, 3, ci, co, device='cuda') * 0.05
        b = torch.randn(co, device='cuda')
        dy = torch.randn(D, D, D, co, device='cuda')
        y = torch.empty(D, D, D, co, device='cuda')
        dx = torch.empty(D, D, D, ci, device='cuda')
        dw = torch.zeros_like(w)
        t = {}
        for mode in (0, 1):
            set_split(mode)
            wp, wpd = ops.pack_conv_weights(w, shape, 0), ops.pack_conv_weights(w, shape, 1)
            t[mode] = (timeit(lambda: ops.conv3d(x, wp, b, co, 1, out=y), reps),
                       timeit(lambda: ops.conv3d_add(dy, wpd, None, x, ci, 2, out=dx), reps) if ci == co else
                       timeit(lambda: ops.conv3d(dy, wpd, None, ci, 0, out=dx), reps),
                       timeit(lambda: ops.conv3d_wgrad(x, dy, dw), reps))
        set_split(0)
        fl = 2.0 * 27 * ci * co * D ** 3 / 1e9
        print('%4d^3 %4d->%-4d ' % (D, ci, co) + ' '.join('%5.3f->%5.3f' % (t[0][k], t[1][k]) for k in range(3)) +
              '   split TF: ' + ' '.join('%6.1f' % (fl / t[1][k]) for k in range(3)))


def folded(reps):
    """the up-sampled channel range of a folded decoder conv (ops.conv3d_up / conv3d_up_dgrad / conv3d_up_wgrad): split vs fp32
    MFMA, error against a float64 evaluation of conv3(UpSampling3D(2)(lo)) at a small size, time at the U-Net's sizes"""
    torch.manual_seed(1)
    print('folded decoder conv, up-sampled channels: error vs float64 (max / rms of the result rms), fp32 MFMA | split')
    for D, cs, cl, co in [(48, 24, 48, 24), (40, 48, 96, 48)]:
        lo_shape = (D, D, D)
        lo = torch.randn(D, D, D, cl, device='cuda')
        dz = torch.randn(2 * D, 2 * D, 2 * D, co, device='cuda')
        w = torch.randn(3, 3, 3, cs + cl, co, device='cuda') * 0.05
        up = lo.double().cpu().repeat_interleave(2, 0).repeat_interleave(2, 1).repeat_interleave(2, 2).requires_grad_(True)
        wu = w[:, :, :, cs:].double().cpu().requires_grad_(True)
        y = F.conv3d(up.permute(3, 0, 1, 2)[None], wu.permute(4, 3, 0, 1, 2), None, padding=1)[0].permute(1, 2, 3, 0)
        y.backward(dz.double().cpu())
        dlo_ref = up.grad.reshape(D, 2, D, 2, D, 2, cl).sum((1, 3, 5))
        res = {}
        for mode in (0, 1):
            set_split(mode)
            wp_u = ops.pack_conv_weights_ex(w, lo_shape, cs, cl, 0, up=True)
            wpd_u = ops.pack_conv_weights_ex(w, lo_shape, cs, cl, 1, up=True)
            yu = ops.conv3d_up(lo, wp_u, None, None, co, 0)
            dlo = ops.conv3d_up_dgrad(dz, wpd_u, cl)
            dW = torch.zeros_like(w)
            dwc = torch.zeros(8, 27, cl, co, device='cuda')
            ops.conv3d_up_wgrad(lo, dz, dwc, dW, cs)
            res[mode] = (errs(yu, y.detach()), errs(dlo, dlo_ref), errs(dW[:, :, :, cs:], wu.grad))
        set_split(0)
        for k, nm in enumerate(('up fwd', 'up dgrad', 'up wgrad')):
            print('lo %2d^3 %3d->%-3d %-9s %.3e / %.3e | %.3e / %.3e' % (D, cl, co, nm, res[0][k][0], res[0][k][1], res[1][k][0], res[1][k][1]))
    print('%-22s %12s %12s %12s   (ms: fp32 MFMA -> split)' % ('layer (low-res grid)', 'up fwd', 'up dgrad', 'up wgrad'))
    for D, cs, cl, co in [(80, 24, 48, 24), (40, 48, 96, 48), (20, 96, 192, 96)]:
        lo_shape = (D, D, D)
        lo = torch.randn(D, D, D, cl, device='cuda')
        dz = torch.randn(2 * D, 2 * D, 2 * D, co, device='cuda')
        w = torch.randn(3, 3, 3, cs + cl, co, device='cuda') * 0.05
        out = torch.empty(2 * D, 2 * D, 2 * D, co, device='cuda')
        dlo = torch.empty(D, D, D, cl, device='cuda')
        dW = torch.zeros_like(w)
        dwc = torch.zeros(8, 27, cl, co, device='cuda')
        t = {}
        for mode in (0, 1):
            set_split(mode)
            wp_u = ops.pack_conv_weights_ex(w, lo_shape, cs, cl, 0, up=True)
            wpd_u = ops.pack_conv_weights_ex(w, lo_shape, cs, cl, 1, up=True)
            t[mode] = (timeit(lambda: ops.conv3d_up(lo, wp_u, None, None, co, 0, out=out), reps),
                       timeit(lambda: ops.conv3d_up_dgrad(dz, wpd_u, cl, out=dlo), reps),
                       timeit(lambda: ops.conv3d_up_wgrad(lo, dz, dwc, dW, cs), reps))
        set_split(0)
        print('%4d^3 %4d->%-4d      ' % (D, cl, co) + ' '.join('%5.3f->%5.3f' % (t[0][k], t[1][k]) for k in range(3)))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--acc', action='store_true')
    ap.add_argument('--time', action='store_true')
    ap.add_argument('--folded', action='store_true', help='only the folded decoder conv comparison')
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--only', default='', help='comma-separated D_cin_cout of the timing table')
    a = ap.parse_args()
    if a.folded:
        folded(a.reps)
        sys.exit(0)
    if a.acc or not a.time:
        accuracy()
    if a.time or not a.acc:
        timing(a.reps, a.only)
