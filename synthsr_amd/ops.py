"""Thin tensor-level wrappers over the C ABI (include/synthsr_hip.h) for the U-Net kernels.

Tensors are contiguous float32 device tensors in NDHWC order ([d0,d1,d2,C]; batch handled by the
caller).  No autograd, no CPU fallback: each function is one (or a few) kernel launches on the
current stream.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib

BN_EPS = 1e-3  # keras.layers.BatchNormalization default epsilon (Keras 2.3.1)

# optional per-launch timing of the conv kernels (bench.py): list of (kind, shape, Cin, Cout, start_evt, end_evt)
_prof = None


# The context handed to every fp32 conv entry point (include/synthsr_hip.h: synthsr_conv_ctx).  The LIBRARY holds no arithmetic
# state and owns no scratch: this module keeps the default arithmetic for the host code that does not pass its own (unet.py,
# critic.py, the tests), one context PER (device, stream) -- each with its own torch-allocated workspace, so that two streams of
# one device never share scratch -- and a counter that moves when the arithmetic is replaced: packed weights are only valid under
# the arithmetic they were packed with (check_layout_epoch).
_arith = 1
_ctx_epoch = 0
_ctx_host = _lib.ConvCtx(arithmetic=1)        # host-only queries (plans, packed sizes): no workspace, usable without a GPU
_ctxs = {}                                    # (device, stream handle) -> (ConvCtx, its workspace tensor)
_ctx_last = (None, None)


def conv_ctx_host():
    """ctypes reference to a workspace-less context of the current arithmetic (host-only entry points)"""
    return ctypes.byref(_ctx_host)


def conv_ctx():
    """ctypes reference to the conv context of the CURRENT device and stream (created on first use: the current arithmetic and
    a workspace of synthsr_conv_workspace_bytes() that torch allocates)"""
    global _ctx_last
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    if _ctx_last[0] == key:
        return _ctx_last[1]
    e = _ctxs.get(key)
    if e is None:
        n = int(_L().synthsr_conv_workspace_bytes())
        ws = torch.empty(n, dtype=torch.uint8, device='cuda:%d' % key[0])
        e = (_lib.ConvCtx(arithmetic=_arith, workspace=ws.data_ptr(), workspace_bytes=n), ws)
        _ctxs[key] = e
    _ctx_last = (key, ctypes.byref(e[0]))
    return _ctx_last[1]


def set_conv_arithmetic(name):
    """replaces this module's default conv arithmetic: 'split' (default) = fp32 convolutions on the bf16 matrix cores through three
    bf16 pieces per operand and six exact partial products (fp32 accumulation, as accurate as the fp32 matrix instructions:
    tests/test_split_gpu.py); 'split9' = the same with all nine partial products (every fp32 product reproduced exactly, 1.5x
    the matrix instructions); 'fp32_mfma' = fp32 matrix instructions everywhere.  Networks must re-pack their weights
    (`repack()`) before their next forward / backward: conv_layout_epoch moves and check_layout_epoch raises on stale packed
    weights.  Returns the previous setting."""
    global _arith, _ctx_epoch, _ctx_host, _ctx_last
    if name not in _lib.CONV_ARITHMETICS:
        raise ValueError('conv arithmetic should be one of %s' % (_lib.CONV_ARITHMETICS,))
    prev = conv_arithmetic()
    if name != prev:
        _arith = _lib.CONV_ARITHMETICS.index(name)
        _ctx_host = _lib.ConvCtx(arithmetic=_arith)
        for key, (c, ws) in list(_ctxs.items()):   # fresh structs (a launch in flight has read its own already): same workspaces
            _ctxs[key] = (_lib.ConvCtx(arithmetic=_arith, workspace=ws.data_ptr(), workspace_bytes=ws.numel()), ws)
        _ctx_last = (None, None)
        _ctx_epoch += 1
    return prev


def check_layout_epoch(epoch, what):
    """a network's packed weights were laid out under conv_layout_epoch() == epoch: using them under another arithmetic would
    plan for one layout and read another (silently wrong results on the Cout = 24 layers) -- raise instead"""
    if epoch is not None and epoch != _ctx_epoch:
        raise RuntimeError('%s: the conv arithmetic changed (ops.set_conv_arithmetic) since these weights were packed; call '
                           'repack() first' % what)


def conv_arithmetic():
    return _lib.CONV_ARITHMETICS[_arith]


def conv_layout_epoch():
    """moves whenever the default conv context was replaced: packed weights of an older epoch are stale -- their size or
    fragment order may belong to another plan"""
    return _ctx_epoch


def conv_runs_split(kind, shape, cin, cout):
    """whether a conv launch of this kind ('conv3d_fwd' | 'conv3d_dgrad' | 'conv3d_wgrad' | 'conv3d_up_fwd' | 'conv3d_up_dgrad',
    the names of the profile records) runs on the split kernels under the CURRENT arithmetic (mirrors the dispatcher: csrc/conv3d.hip plan_fwd /
    dispatch_wgrad); used by the benchmarks to price a kernel against the right peak"""
    kinds = ('conv3d_fwd', 'conv3d_dgrad', 'conv3d_wgrad', 'conv3d_up_fwd', 'conv3d_up_dgrad', 'conv3d_up_wgrad')
    if conv_arithmetic() == 'fp32_mfma' or kind not in kinds:
        return False
    d0, d1, d2 = [int(v) for v in shape[:3]]
    if kind == 'conv3d_up_wgrad':   # (low-res shape, Cl, Cout): csrc/conv3d.hip up_wgrad_takes_split
        rc = int(_L().synthsr_conv3d_up_wgrad_runs_split(conv_ctx_host(), _lib.i3((d0, d1, d2)), int(cin), int(cout)))
        if rc < 0:
            _lib.check(rc, 'conv3d_up_wgrad_runs_split')
        return rc == 1
    if kind == 'conv3d_wgrad':   # the dispatcher's own condition (csrc/conv3d.hip: wgrad_takes_split)
        rc = int(_L().synthsr_conv3d_wgrad_runs_split(conv_ctx_host(), _lib.i3((d0, d1, d2)), int(cin), int(cout)))
        if rc < 0:
            _lib.check(rc, 'conv3d_wgrad_runs_split')
        return rc == 1
    # folded decoder convs are recorded with (low-res shape, Cl, Cout): forward = plan kind 2 on (Cl -> Cout), data gradient =
    # plan kind 0 on (Cout -> Cl)
    plan_kind, ce, co = {'conv3d_up_fwd': (2, cin, cout), 'conv3d_up_dgrad': (0, cout, cin)}.get(kind, (1, cin, cout))
    out = (ctypes.c_int64 * 8)()
    _lib.check(_L().synthsr_conv3d_plan(conv_ctx_host(), _lib.i3((d0, d1, d2)), int(ce), int(co), plan_kind, out), 'conv3d_plan')
    return int(out[2]) <= -100


def set_deterministic(on=True):
    """process-wide switch (include/synthsr_hip_tuning.h: synthsr_set_deterministic): bit-identical results run after run on
    the same inputs -- every cross-workgroup float accumulation happens in a fixed order, no split-K forward.  Scope: the
    whole PROCESS on the CURRENT device (every network, critic and predictor is switched together) and ONE stream (kernels
    of two streams must not overlap while it is on).  Returns the previous setting."""
    global _deterministic
    prev = _deterministic
    torch.cuda.synchronize()
    _lib.check(_L().synthsr_set_deterministic(int(bool(on))), 'set_deterministic')
    _deterministic = bool(on)
    if on:
        _register_det_planes(64 << 20)
    return prev


_det_planes = {}    # device -> the torch buffer registered as the private dW planes of the deterministic weight-gradient flush


def _register_det_planes(min_bytes):
    """the planes are CALLER memory (include/synthsr_hip_tuning.h: synthsr_set_deterministic_workspace): torch allocates, grown to
    1.25x the library's recorded demand whenever a weight gradient reported SYNTHSR_EWORKSPACE"""
    dev = torch.cuda.current_device()
    want = max(int(min_bytes), int(1.25 * int(_L().synthsr_deterministic_workspace_demand())))
    cur = _det_planes.get(dev)
    if cur is None or cur.numel() < want:
        torch.cuda.synchronize()
        _det_planes[dev] = None     # release the old buffer before asking for the larger one
        cur = torch.empty(want, dtype=torch.uint8, device='cuda:%d' % dev)
        _det_planes[dev] = cur
        _lib.check(_L().synthsr_set_deterministic_workspace(_lib.ptr(cur), cur.numel()), 'set_deterministic_workspace')


def _check_wgrad(call, what):
    """weight-gradient launches: in deterministic mode the call may report that the registered plane buffer is too small
    (nothing was launched) -- register a larger one and repeat ONCE"""
    rc = call()
    if rc == -3 and _deterministic:
        _register_det_planes(0)
        rc = call()
    _lib.check(rc, what)


def deterministic_status():
    """0 off; 1 on and every ordered flush completed in order; 2 on but a wait timed out (order not guaranteed)"""
    torch.cuda.synchronize()
    return int(_L().synthsr_deterministic_status())


_deterministic = False


def profile_start():
    global _prof
    _prof = []


def profile_stop():
    global _prof, _prof_paused
    p = _prof if _prof is not None else _prof_paused
    _prof, _prof_paused = None, None
    return p


_prof_paused = None


def profile_pause():
    """stop recording events (they cost ~8 us per launch) but keep what was collected for profile_stop()"""
    global _prof, _prof_paused
    if _prof is not None:
        _prof_paused, _prof = _prof, None


# ---- roctx ranges (tracing): named CPU-side ranges around the phases of a step and around every conv /
# generator launch, visible in `rocprofv3 --marker-trace`.  Off unless SYNTHSR_ROCTX=1 (one ctypes call per range otherwise).
_roctx = None


def _roctx_lib():
    global _roctx
    if _roctx is None:
        _roctx = False
        if os.environ.get('SYNTHSR_ROCTX') == '1':
            import ctypes
            for name in ('libroctx64.so', 'librocprofiler-sdk-roctx.so'):
                try:
                    lib = ctypes.CDLL(name)
                    lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    lib.roctxRangePushA.restype = ctypes.c_int
                    lib.roctxRangePop.restype = ctypes.c_int
                    _roctx = lib
                    break
                except (OSError, AttributeError):
                    continue
    return _roctx


class trace_range:
    """`with ops.trace_range('backward'):` -- a roctx range when SYNTHSR_ROCTX=1 and the marker library is present, else nothing"""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        lib = _roctx_lib()
        self.on = bool(lib)
        if self.on:
            lib.roctxRangePushA(str(self.name).encode())
        return self

    def __exit__(self, *a):
        if self.on:
            _roctx.roctxRangePop()


def timed(kind, shape=(0, 0, 0), cin=0, cout=0):
    """context manager: HIP events around a region on the launch stream, recorded while profile_start() is active
    (bench.py: conv launches and the generator's kernels); free otherwise"""
    return _Timed(kind, shape, cin, cout)


class _Timed:
    def __init__(self, kind, shape, cin, cout):
        self.meta = (kind, tuple(int(s) for s in shape), int(cin), int(cout))

    def __enter__(self):
        self.rng = None
        if _roctx_lib():
            k, sh, ci, co = self.meta
            self.rng = trace_range('%s %dx%dx%d %d->%d' % (k, sh[0], sh[1], sh[2], ci, co)).__enter__()
        if _prof is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        if _prof is not None:
            self.e.record()
            _prof.append(self.meta + (self.s, self.e))
        if self.rng is not None:
            self.rng.__exit__()


def _L():
    return _lib.load()


def _sym(name, t):
    """entry point `name` for float32 tensors, `name_bf16` for bfloat16 ones (same arguments)"""
    return getattr(_L(), name + '_bf16' if t.dtype == torch.bfloat16 else name)


def pack_conv_weights(w, shape, mode=0, out=None):
    """w: Keras Conv3D kernel [3,3,3,Cin,Cout] -> MFMA fragment order for a layer of spatial size `shape`
    (mode 0 fwd, 1 data-gradient)"""
    lib = _L()
    Cin, Cout = int(w.shape[3]), int(w.shape[4])
    s3 = _lib.i3(shape[:3])
    n = lib.synthsr_conv3d_pack(conv_ctx_host(), None, None, s3, Cin, Cout, mode, None)
    if n < 0:
        _lib.check(int(n), 'conv3d_pack(size)')
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=w.device)
    assert out.numel() == n
    r = lib.synthsr_conv3d_pack(conv_ctx(), _lib.ptr(w), _lib.ptr(out), s3, Cin, Cout, mode, _lib.stream())
    if r < 0:
        _lib.check(int(r), 'conv3d_pack')
    return out


def pack_conv_weights_ex(w, shape, ci_off, cin, mode=0, up=False, out=None):
    """pack the input-channel range [ci_off, ci_off+cin) of a Keras kernel w [3,3,3,Cin_total,Cout]; `up`: the 8 parity
    weight sets of the nearest-upsample folding (shape = LOW-RES spatial shape); up = 2: those of a stride-2 conv"""
    lib = _L()
    cin_total, cout = int(w.shape[3]), int(w.shape[4])
    s3 = _lib.i3(shape[:3])
    n = lib.synthsr_conv3d_pack_ex(conv_ctx_host(), None, None, s3, cin_total, int(ci_off), int(cin), cout, mode, int(up), None)
    if n < 0:
        _lib.check(int(n), 'conv3d_pack_ex(size)')
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=w.device)
    assert out.numel() == n
    r = lib.synthsr_conv3d_pack_ex(conv_ctx(), _lib.ptr(w), _lib.ptr(out), s3, cin_total, int(ci_off), int(cin), cout, mode,
                                   int(up), _lib.stream())
    if r < 0:
        _lib.check(int(r), 'conv3d_pack_ex')
    return out


def conv3d_up(lo, wpacked8, bias, addend, Cout, act=1, out=None):
    """act(conv3(UpSampling3D(2)(lo)) + addend + bias) evaluated on the low-res tensor (8 parity convs); bf16: the raw
    partial sums only (bias / addend / activation come with the skip-channel conv, conv3d_add)"""
    lib = _L()
    s = lo.shape
    if lo.dtype == torch.bfloat16:
        assert bias is None and addend is None and act == 0
        if out is None:
            out = torch.empty((2 * s[0], 2 * s[1], 2 * s[2], Cout), dtype=torch.bfloat16, device=lo.device)
        with _Timed('conv3d_bf16_up_fwd', s[:3], s[3], Cout):
            _lib.check(lib.synthsr_conv3d_bf16_up_fwd(_lib.ptr(lo), _lib.ptr(wpacked8), _lib.ptr(out), _lib.i3(s[:3]),
                                                      int(s[3]), int(Cout), _lib.stream()), 'conv3d_bf16_up_fwd')
        return out
    if out is None:
        out = torch.empty((2 * s[0], 2 * s[1], 2 * s[2], Cout), dtype=torch.float32, device=lo.device)
    with _Timed('conv3d_up_fwd', s[:3], s[3], Cout):
        _lib.check(lib.synthsr_conv3d_up_fwd(conv_ctx(), _lib.ptr(lo), _lib.ptr(wpacked8), _lib.ptr(bias), _lib.ptr(addend),
                                             _lib.ptr(out), _lib.i3(s[:3]), int(s[3]), int(Cout), int(act),
                                             _lib.stream()), 'conv3d_up_fwd')
    return out


def conv3d_up_dgrad(dout, wpacked8, Cl, out=None):
    """gradient w.r.t. the low-res tensor of conv3d_up; dout [2*lo_shape, Cout]"""
    lib = _L()
    s = dout.shape
    lo_shape = (s[0] // 2, s[1] // 2, s[2] // 2)
    if dout.dtype == torch.bfloat16:
        if out is None:
            out = torch.empty(lo_shape + (Cl,), dtype=torch.bfloat16, device=dout.device)
        scratch = _bf16_scratch.get(dout.device)   # fp32 partial planes of the split-K path (small levels)
        need = 8 * lo_shape[0] * lo_shape[1] * lo_shape[2] * int(Cl) if lo_shape[0] * lo_shape[1] * lo_shape[2] <= 32 ** 3 else 0
        if scratch is None or scratch.numel() < need:
            scratch = _bf16_scratch[dout.device] = torch.empty(max(need, 1 << 20), dtype=torch.float32, device=dout.device)
        with _Timed('conv3d_bf16_up_dgrad', lo_shape, Cl, s[3]):
            _lib.check(lib.synthsr_conv3d_bf16_up_dgrad(_lib.ptr(dout), _lib.ptr(wpacked8), _lib.ptr(out), _lib.i3(lo_shape),
                                                        int(Cl), int(s[3]), _lib.ptr(scratch), scratch.numel(),
                                                        _lib.stream()), 'conv3d_bf16_up_dgrad')
        return out
    if out is None:
        out = torch.empty(lo_shape + (Cl,), dtype=torch.float32, device=dout.device)
    with _Timed('conv3d_up_dgrad', lo_shape, Cl, s[3]):
        _lib.check(lib.synthsr_conv3d_up_dgrad(conv_ctx(), _lib.ptr(dout), _lib.ptr(wpacked8), _lib.ptr(out), _lib.i3(lo_shape),
                                               int(Cl), int(s[3]), _lib.stream()), 'conv3d_up_dgrad')
    return out


def conv3d_up_wgrad(lo, dout, dwc, dw, ci_off, dwc_is_zero=False):
    """weight gradient of the up-sampled channel range: dwc [8,27,Cl,Cout] scratch, dw (+=).  dwc is zeroed here unless the
    caller vouches that it is all zeros (`dwc_is_zero`): the unpack kernel CONSUMES the partials and leaves zeros behind, so a
    persistent scratch buffer needs the memset once, not once per call (UNet3D keeps that book)"""
    lib = _L()
    s = lo.shape
    if not dwc_is_zero:
        dwc.zero_()
    if lo.dtype == torch.bfloat16:
        with _Timed('conv3d_bf16_up_wgrad', s[:3], s[3], dout.shape[3]):
            _check_wgrad(lambda: lib.synthsr_conv3d_bf16_up_wgrad(_lib.ptr(lo), _lib.ptr(dout), _lib.ptr(dwc), _lib.i3(s[:3]), int(s[3]),
                                                        int(dout.shape[3]), _lib.stream()), 'conv3d_bf16_up_wgrad')
        _lib.check(lib.synthsr_conv3d_up_unpack(_lib.ptr(dwc), _lib.ptr(dw), int(dw.shape[3]), int(ci_off), int(s[3]),
                                                int(dout.shape[3]), _lib.stream()), 'conv3d_up_unpack')
        return dw
    with _Timed('conv3d_up_wgrad', s[:3], s[3], dout.shape[3]):
        _check_wgrad(lambda: lib.synthsr_conv3d_up_wgrad(conv_ctx(), _lib.ptr(lo), _lib.ptr(dout), _lib.ptr(dwc), _lib.i3(s[:3]),
                                               int(s[3]), int(dout.shape[3]), _lib.stream()), 'conv3d_up_wgrad')
    _lib.check(lib.synthsr_conv3d_up_unpack(_lib.ptr(dwc), _lib.ptr(dw), int(dw.shape[3]), int(ci_off), int(s[3]),
                                            int(dout.shape[3]), _lib.stream()), 'conv3d_up_unpack')
    return dw


def conv3d_wgrad_part(x, dout, dw, ci_off, dbias=None):
    """dw [3,3,3,Cin_total,Cout] += gradient of the input-channel range [ci_off, ci_off + x.shape[3]);
    dbias [Cout] (optional) += sum over voxels of dout"""
    lib = _L()
    s = x.shape
    if x.dtype == torch.bfloat16:
        with _Timed('conv3d_bf16_wgrad', s[:3], s[3], dout.shape[3]):
            _check_wgrad(lambda: lib.synthsr_conv3d_bf16_wgrad_part(_lib.ptr(x), _lib.ptr(dout), _lib.ptr(dw), _lib.ptr(dbias),
                                                          _lib.i3(s[:3]), int(dw.shape[3]), int(ci_off), int(s[3]),
                                                          int(dout.shape[3]), _lib.stream()), 'conv3d_bf16_wgrad_part')
        return dw
    with _Timed('conv3d_wgrad', s[:3], s[3], dout.shape[3]):
        _check_wgrad(lambda: lib.synthsr_conv3d_wgrad_bias(conv_ctx(), _lib.ptr(x), _lib.ptr(dout), _lib.ptr(dw), _lib.ptr(dbias),
                                                 _lib.i3(s[:3]), int(dw.shape[3]), int(ci_off), int(s[3]),
                                                 int(dout.shape[3]), _lib.stream()), 'conv3d_wgrad_bias')
    return dw


def conv3d(x, wpacked, bias, Cout, act=1, out=None):
    """x [d0,d1,d2,Cin] -> [d0,d1,d2,Cout]; act: 0 linear, 1 ELU"""
    if x.dtype == torch.bfloat16:
        return conv3d_bf16(x, wpacked, bias, Cout, act, out=out)
    lib = _L()
    s = x.shape
    if out is None:
        out = torch.empty((s[0], s[1], s[2], Cout), dtype=torch.float32, device=x.device)
    with _Timed('conv3d_fwd' if bias is not None else 'conv3d_dgrad', s[:3], s[3], Cout):
        _lib.check(lib.synthsr_conv3d_fwd(conv_ctx(), _lib.ptr(x), _lib.ptr(wpacked), _lib.ptr(bias), _lib.ptr(out),
                                          _lib.i3(s[:3]), int(s[3]), int(Cout), int(act), _lib.stream()), 'conv3d_fwd')
    return out


def conv3d_stats(x, wpacked, bias, Cout, stats, ws, act=1, out=None):
    """conv3d + BatchNorm batch statistics of its output (fused into the conv epilogue where the kernel supports it)"""
    if x.dtype == torch.bfloat16:
        return conv3d_bf16(x, wpacked, bias, Cout, act, stats=stats, out=out)
    lib = _L()
    s = x.shape
    if out is None:
        out = torch.empty((s[0], s[1], s[2], Cout), dtype=torch.float32, device=x.device)
    with _Timed('conv3d_fwd', s[:3], s[3], Cout):
        _lib.check(lib.synthsr_conv3d_fwd_stats(conv_ctx(), _lib.ptr(x), _lib.ptr(wpacked), _lib.ptr(bias), _lib.ptr(out),
                                                _lib.i3(s[:3]), int(s[3]), int(Cout), int(act), _lib.ptr(stats),
                                                _lib.ptr(ws), _lib.stream()), 'conv3d_fwd_stats')
    return out


def conv3d_add(x, wpacked, bias, addend, Cout, act=1, out=None):
    """act 0/1: act(conv3(x) + addend + bias), `addend` may be `out` itself (in-place accumulation);
    act 2: conv3(x) * elu'(addend) -- data gradient fused with the ELU backward of the layer that produced `addend`"""
    if x.dtype == torch.bfloat16:
        if act == 0:
            raise NotImplementedError('bf16: addend epilogues are ELU(conv + bias + addend) (act 1) and conv * ELU\'(addend) (act 2)')
        return conv3d_bf16(x, wpacked, bias, Cout, 5 if act == 1 else 2, below=addend, out=out)
    lib = _L()
    s = x.shape
    if out is None:
        out = torch.empty((s[0], s[1], s[2], Cout), dtype=torch.float32, device=x.device)
    with _Timed('conv3d_dgrad' if act == 2 else 'conv3d_fwd', s[:3], s[3], Cout):
        _lib.check(lib.synthsr_conv3d_fwd_add(conv_ctx(), _lib.ptr(x), _lib.ptr(wpacked), _lib.ptr(bias), _lib.ptr(addend),
                                              _lib.ptr(out), _lib.i3(s[:3]), int(s[3]), int(Cout), int(act),
                                              _lib.stream()), 'conv3d_fwd_add')
    return out


def conv3d_wgrad(x, dout, dw, dbias=None):
    """dw [3,3,3,Cin,Cout] += sum_v x[v+t-1] (x) dout[v]; dbias [Cout] (optional) += sum_v dout[v]"""
    if x.dtype == torch.bfloat16:
        return conv3d_wgrad_bf16(x, dout, dw, dbias)
    return conv3d_wgrad_part(x, dout, dw, 0, dbias)


def elu_bwd(dy, y, dy2=None, dbias=None, out=None):
    lib = _L()
    C = int(y.shape[-1])
    nvox = y.numel() // C
    if out is None:
        out = torch.empty_like(y)
    _lib.check(_sym('synthsr_elu_bwd', y)(_lib.ptr(dy), _lib.ptr(dy2), _lib.ptr(y), _lib.ptr(out), _lib.ptr(dbias), nvox, C,
                                   _lib.stream()), 'elu_bwd')
    return out


def scale_channels(x, scale, out=None):
    """out[b, ..., c] = x[b, ..., c] * scale[b, c]: the dropped-out tensor of a batch with one feature mask per sample
    (KL.Dropout(noise_shape=[None, 1, 1, 1, C]), ext/neuron/models.py:320-324); x: the B volumes stacked along the first
    axis ([B * d0, d1, d2, C], fp32 or bf16; scale float32); in place allowed"""
    C = int(x.shape[-1])
    B = int(scale.shape[0])
    if x.dtype not in (torch.float32, torch.bfloat16) or scale.dtype != torch.float32 or tuple(scale.shape) != (B, C) or \
            (x.numel() // C) % B:
        raise ValueError('scale_channels: B volumes stacked along the first axis, and a float32 scale [B, C]')
    if out is None:
        out = torch.empty_like(x)
    nvox = x.numel() // C
    _lib.check(_sym('synthsr_scale_channels', x)(_lib.ptr(x), _lib.ptr(out), nvox, C, _lib.ptr(scale), nvox // B, _lib.stream()),
               'scale_channels')
    return out


def elu_bwd_drop(dy, y, drop, dy2=None, dbias=None, out=None, bn=None, head=None, eps=BN_EPS):
    """ELU backward of a conv output y ([B * d0, d1, d2, C]) whose consumer read drop[b, c] * y (see synthsr_elu_bwd_drop);
    bn = (stats, gamma, sums) when that consumer is a BatchNorm, head = (dpred, whead) for the rank-1 gradient of the head"""
    C = int(y.shape[-1])
    B = int(drop.shape[0])
    if out is None:
        out = torch.empty_like(y)
    stats, gamma, sums = bn if bn is not None else (None, None, None)
    dpred, whead = head if head is not None else (None, None)
    nvox = y.numel() // C
    _lib.check(_sym('synthsr_elu_bwd_drop', y)(_lib.ptr(dy), _lib.ptr(dy2), _lib.ptr(y), _lib.ptr(out), _lib.ptr(dbias), nvox, C,
                                        _lib.ptr(stats), _lib.ptr(gamma), eps, _lib.ptr(sums), _lib.ptr(dpred), _lib.ptr(whead),
                                        _lib.ptr(drop), nvox // B, _lib.stream()), 'elu_bwd_drop')
    return out


def bn_reduce_bwd(dy, x, stats, sums, eps=BN_EPS):
    """pass 1 of the BN backward: sums [2C] (zeroed by caller) += [sum dy, sum dy*xhat]"""
    lib = _L()
    C = int(x.shape[-1])
    _lib.check(_sym('synthsr_bn_bwd_reduce', x)(_lib.ptr(dy), _lib.ptr(x), x.numel() // C, C, _lib.ptr(stats), eps,
                                         _lib.ptr(sums), _lib.stream()), 'bn_bwd_reduce')
    return sums


def bn_elu_bwd(dy, y, stats, gamma, sums, dy2=None, dbias=None, out=None, eps=BN_EPS):
    """fused pass 2 of the BN backward + ELU backward (dy = gradient w.r.t. BN(y))"""
    lib = _L()
    C = int(y.shape[-1])
    if out is None:
        out = torch.empty_like(y)
    _lib.check(_sym('synthsr_bn_elu_bwd', y)(_lib.ptr(dy), _lib.ptr(dy2), _lib.ptr(y), _lib.ptr(out), _lib.ptr(dbias),
                                      y.numel() // C, C, _lib.ptr(stats), _lib.ptr(gamma), eps, _lib.ptr(sums),
                                      _lib.stream()), 'bn_elu_bwd')
    return out


def bn_pool_elu_bwd(dpool, y, stats, gamma, beta, sums, dy2=None, dbias=None, out=None, eps=BN_EPS):
    """MaxPooling3D backward + BN backward (pass 2) + ELU backward of an encoder level in one pass: dpool = gradient w.r.t.
    maxpool(BN(y)), sums from bn_maxpool_bwd(..., out=False, sums=sums); bit-identical to bn_maxpool_bwd + bn_elu_bwd"""
    s = y.shape
    if out is None:
        out = torch.empty_like(y)
    _lib.check(_sym('synthsr_bn_pool_elu_bwd', y)(_lib.ptr(dpool), _lib.ptr(y), _lib.ptr(dy2), _lib.ptr(out), _lib.ptr(dbias),
                                                  _lib.i3(s[:3]), int(s[3]), _lib.ptr(stats), _lib.ptr(gamma), _lib.ptr(beta),
                                                  _lib.ptr(sums), eps, _lib.stream()), 'bn_pool_elu_bwd')
    return out


def bn_stats(x, stats, ws):
    lib = _L()
    C = int(x.shape[-1])
    _lib.check(_sym('synthsr_bn_stats', x)(_lib.ptr(x), x.numel() // C, C, _lib.ptr(stats), _lib.ptr(ws), _lib.stream()),
               'bn_stats')
    return stats


def bn_apply(x, stats, gamma, beta, out=None, eps=BN_EPS):
    lib = _L()
    C = int(x.shape[-1])
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_sym('synthsr_bn_apply', x)(_lib.ptr(x), _lib.ptr(out), x.numel() // C, C, _lib.ptr(stats), _lib.ptr(gamma),
                                           _lib.ptr(beta), eps, _lib.stream()), 'bn_apply')
    return out


def bn_maxpool(x, stats, gamma, beta, out=None, eps=BN_EPS):
    lib = _L()
    s = x.shape
    if out is None:
        out = torch.empty((s[0] // 2, s[1] // 2, s[2] // 2, s[3]), dtype=x.dtype, device=x.device)
    _lib.check(_sym('synthsr_bn_maxpool', x)(_lib.ptr(x), _lib.ptr(out), _lib.i3(s[:3]), int(s[3]), _lib.ptr(stats),
                                      _lib.ptr(gamma), _lib.ptr(beta), eps, _lib.stream()), 'bn_maxpool')
    return out


def bn_maxpool_bwd(dy, x, stats, gamma, beta, out=None, eps=BN_EPS, sums=None):
    """gradient of maxpool(BN(x)) w.r.t. BN(x); sums [2C] (optional, += ) = bn_reduce_bwd(result, x) for free;
    out=False (with sums): the sums only, the routed gradient is not written (bn_pool_elu_bwd re-derives it)"""
    lib = _L()
    s = x.shape
    if out is False:
        assert sums is not None
        out = None
    elif out is None:
        out = torch.empty_like(x)
    _lib.check(_sym('synthsr_bn_maxpool_bwd_ex', x)(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(out), _lib.i3(s[:3]), int(s[3]),
                                             _lib.ptr(stats), _lib.ptr(gamma), _lib.ptr(beta), eps, _lib.ptr(sums),
                                             _lib.stream()), 'bn_maxpool_bwd')
    return out


def bn_bwd(dy, x, stats, gamma, sums, out=None, eps=BN_EPS):
    """sums [2C] (zeroed by caller) receives [sum dy (=dbeta), sum dy*xhat (=dgamma)]; returns dx"""
    lib = _L()
    C = int(x.shape[-1])
    nvox = x.numel() // C
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_sym('synthsr_bn_bwd_reduce', x)(_lib.ptr(dy), _lib.ptr(x), nvox, C, _lib.ptr(stats), eps, _lib.ptr(sums),
                                         _lib.stream()), 'bn_bwd_reduce')
    _lib.check(lib.synthsr_bn_bwd_apply(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(out), nvox, C, _lib.ptr(stats),
                                        _lib.ptr(gamma), eps, _lib.ptr(sums), _lib.stream()), 'bn_bwd_apply')
    return out


def upsample_concat(skip, lo, stats, gamma, beta, out=None, eps=BN_EPS):
    lib = _L()
    s = skip.shape
    Cs, Cl = int(s[3]), int(lo.shape[3])
    if out is None:
        out = torch.empty((s[0], s[1], s[2], Cs + Cl), dtype=skip.dtype, device=skip.device)
    _lib.check(_sym('synthsr_upsample_concat', skip)(_lib.ptr(skip), _lib.ptr(lo), _lib.ptr(out), _lib.i3(s[:3]), Cs, Cl,
                                           _lib.ptr(stats), _lib.ptr(gamma), _lib.ptr(beta), eps, _lib.stream()),
               'upsample_concat')
    return out


def upsample_concat_bwd(dcat, Cs, Cl, dskip=None, dlo=None):
    lib = _L()
    s = dcat.shape
    if dskip is None:
        dskip = torch.empty((s[0], s[1], s[2], Cs), dtype=dcat.dtype, device=dcat.device)
    if dlo is None:
        dlo = torch.empty((s[0] // 2, s[1] // 2, s[2] // 2, Cl), dtype=dcat.dtype, device=dcat.device)
    _lib.check(_sym('synthsr_upsample_concat_bwd', dcat)(_lib.ptr(dcat), _lib.ptr(dskip), _lib.ptr(dlo), _lib.i3(s[:3]), Cs, Cl,
                                               _lib.stream()), 'upsample_concat_bwd')
    return dskip, dlo


def head_l1_fwd(x, stats, gamma, beta, w, b, target, loss, pred=None, dpred=None, residual=None, res_stride=1,
                res_off=0, eps=BN_EPS):
    lib = _L()
    C = int(x.shape[-1])
    nvox = x.numel() // C
    _lib.check(lib.synthsr_head_l1_fwd(_lib.ptr(x), nvox, C, _lib.ptr(stats), _lib.ptr(gamma), _lib.ptr(beta), eps,
                                       _lib.ptr(w), _lib.ptr(b), _lib.ptr(residual), int(res_stride), int(res_off),
                                       _lib.ptr(target), _lib.ptr(pred), _lib.ptr(dpred), _lib.ptr(loss),
                                       _lib.stream()), 'head_l1_fwd')
    return loss


LOSS_KINDS = {'l1': 0, 'l2': 1, 'laplace': 2}


def head_loss_fwd(x, stats, gamma, beta, w, b, target, loss, kind='l1', crop=None, pred=None, dpred=None, residual=None,
                  res_stride=1, res_off=0, eps=BN_EPS, ab=None):
    """unet_likelihood + regression loss (metrics_model.py:30-132).  x [d0,d1,d2,C]; w [C,K]; target [nvox*n] for n
    regression targets: K = n ('l1', 'l2') or 2n ('laplace': intensities then spreads); crop = (begin[3], size[3]) of the
    loss_cropping box or None; residual [nvox, res_stride] with res_off = the channel (or one per target) added to the
    intensities; pred / dpred [nvox*K]"""
    lib = _L()
    C = int(x.shape[-1])
    if x.dim() != 4:
        raise ValueError('x should be [d0, d1, d2, C]')
    K = w.numel() // C
    nvox = x.numel() // C
    n = K // 2 if kind == 'laplace' else K
    if w.numel() != C * K or K < 1 or K > 4 or (kind == 'laplace' and K % 2) or target.numel() != nvox * n:
        raise ValueError('head with %d output channels / target of %d values do not fit the %s loss on %d voxels'
                         % (K, target.numel(), kind, nvox))
    shape = _lib.I3(*[int(v) for v in x.shape[:3]])
    box = None
    if crop is not None:
        box = (_lib.c_int * 6)(*([int(v) for v in crop[0]] + [int(v) for v in crop[1]]))
    offs = [int(res_off)] * n if np.ndim(res_off) == 0 else [int(v) for v in res_off]
    if len(offs) != n:
        raise ValueError('one residual channel per regression target is needed (%d given, %d targets)' % (len(offs), n))
    if ab is not None:   # one l1 / l2 target: also the sums head_bwd_from_sums needs, ab [C + 1] (zeroed by the caller)
        if K != 1 or kind == 'laplace' or ab.numel() < C + 1:
            raise ValueError('the fused backward sums exist for the 1-channel l1 / l2 head')
        _lib.check(_sym('synthsr_head_loss_fwd_ab', x)(_lib.ptr(x), shape, C, _lib.ptr(stats), _lib.ptr(gamma), _lib.ptr(beta), eps,
                                                      _lib.ptr(w), _lib.ptr(b), _lib.ptr(residual), int(res_stride), offs[0],
                                                      _lib.ptr(target), _lib.ptr(pred), _lib.ptr(dpred), _lib.ptr(loss),
                                                      LOSS_KINDS[kind], box, _lib.ptr(ab), _lib.stream()), 'head_loss_fwd_ab')
        return loss
    offs = (_lib.c_int * 4)(*(offs + [0] * (4 - n)))
    _lib.check(_sym('synthsr_head_loss_fwd', x)(_lib.ptr(x), shape, C, _lib.ptr(stats), _lib.ptr(gamma), _lib.ptr(beta), eps,
                                         _lib.ptr(w), _lib.ptr(b), K, _lib.ptr(residual), int(res_stride), offs,
                                         _lib.ptr(target), _lib.ptr(pred), _lib.ptr(dpred), _lib.ptr(loss),
                                         LOSS_KINDS[kind], box, _lib.stream()), 'head_loss_fwd')
    return loss


_SSIM_TAPS = None


def ssim_taps():
    """1-D factor of TensorFlow's `_fspecial_gauss(11, 1.5)` window (softmax of -(x^2+y^2)/(2 sigma^2) = outer product
    of the normalised 1-D Gaussians), float32"""
    global _SSIM_TAPS
    if _SSIM_TAPS is None:
        c = np.arange(11, dtype=np.float32) - np.float32(5)
        g = np.exp(np.square(c) * np.float32(-0.5 / (1.5 * 1.5)), dtype=np.float32)
        _SSIM_TAPS = (g / g.sum(dtype=np.float32)).astype(np.float32)
    return _SSIM_TAPS


SSIM_ORIENTATIONS = (((1, 2), 2.0 / 3.0), ((2, 0), 1.0 / 3.0))  # window axes, weight (metrics_model.py:109-125: xy + xz
# are both slices across axis 0 - the xz transpose only swaps the window axes; yz = slices across axis 1)


def ssim_loss(pred, target, shape, loss, dpred, crop=None, scratch=None):
    """loss += -(1/3)(ssim_xy + ssim_xz + ssim_yz) of SynthSR/metrics_model.py:105-125 and dpred = its gradient w.r.t.
    pred (zero outside the loss_cropping box).  pred, target, dpred: [nvox] of the volume `shape`; crop = (begin[3],
    size[3]) or None; scratch(key, numel) -> float32 device buffer (reused between steps)"""
    lib = _L()
    st = _lib.stream()
    shape = [int(v) for v in shape]
    lo, n = ([0, 0, 0], shape) if crop is None else ([int(v) for v in crop[0]], [int(v) for v in crop[1]])
    if any(n[a] < 11 for a in (0, 1, 2)):
        raise ValueError('the SSIM window needs at least 11 voxels per axis, the loss is evaluated on %s' % (n,))
    if scratch is None:
        cache = {}

        def scratch(key, numel):
            if key not in cache or cache[key].numel() < numel:
                cache[key] = torch.empty(numel, dtype=torch.float32, device=pred.device)
            return cache[key]
    nb = n[0] * n[1] * n[2]
    sh3 = _lib.I3(*shape)
    box = None if crop is None else (_lib.c_int * 6)(*(lo + n))
    taps = (ctypes.c_float * 11)(*[float(v) for v in ssim_taps()])
    maps = scratch('ssim_maps', 4 * nb)
    t1 = scratch('ssim_t1', 4 * nb)
    t2 = scratch('ssim_t2', 4 * nb)
    _lib.check(lib.synthsr_ssim_products(_lib.ptr(pred), _lib.ptr(target), sh3, box, _lib.ptr(maps), st), 'ssim_products')
    dpred.zero_()
    for (a, b), weight in SSIM_ORIENTATIONS:
        sa = list(n)
        sa[a] -= 10
        sab = list(sa)
        sab[b] -= 10
        nq = sab[0] * sab[1] * sab[2]
        _lib.check(lib.synthsr_ssim_filter(_lib.ptr(maps), _lib.ptr(t1), _lib.I3(*n), a, 0, 4, taps, st), 'ssim_filter')
        _lib.check(lib.synthsr_ssim_filter(_lib.ptr(t1), _lib.ptr(t2), _lib.I3(*sa), b, 0, 4, taps, st), 'ssim_filter')
        _lib.check(lib.synthsr_ssim_point(_lib.ptr(t2), nq, 1.0, -weight / nq, _lib.ptr(loss), _lib.ptr(t1), st),
                   'ssim_point')
        _lib.check(lib.synthsr_ssim_filter(_lib.ptr(t1), _lib.ptr(t2), _lib.I3(*sab), b, 1, 3, taps, st), 'ssim_filter')
        _lib.check(lib.synthsr_ssim_filter(_lib.ptr(t2), _lib.ptr(t1), _lib.I3(*sa), a, 1, 3, taps, st), 'ssim_filter')
        _lib.check(lib.synthsr_ssim_combine(_lib.ptr(t1), _lib.ptr(pred), _lib.ptr(target), sh3, box, _lib.ptr(dpred), st),
                   'ssim_combine')
    return loss


def head_bwd_from_sums(ab, gamma, beta, w, dw, db, bn_sums=None):
    """head_bwd (dbn not materialised) from the sums head_loss_fwd(..., ab=...) accumulated: dw, db, bn_sums (+=)"""
    C = int(gamma.numel())
    _lib.check(_L().synthsr_head_bwd_from_sums(_lib.ptr(ab), C, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(w), _lib.ptr(dw),
                                               _lib.ptr(db), _lib.ptr(bn_sums), _lib.stream()), 'head_bwd_from_sums')


def head_bwd_multi(dpred, x, stats, gamma, beta, w, dbn, dw, db, eps=BN_EPS):
    """K-channel head backward (2 <= K <= 4): dbn = dpred @ w^T written, dw [C,K] and db [K] accumulated"""
    lib = _L()
    C = int(x.shape[-1])
    nvox = x.numel() // C
    K = w.numel() // C
    _lib.check(_sym('synthsr_head_bwd_multi', x)(_lib.ptr(dpred), _lib.ptr(x), nvox, C, K, _lib.ptr(stats), _lib.ptr(gamma),
                                          _lib.ptr(beta), eps, _lib.ptr(w), _lib.ptr(dbn), _lib.ptr(dw), _lib.ptr(db),
                                          _lib.stream()), 'head_bwd_multi')
    return dbn


def head_bwd(dpred, x, stats, gamma, beta, w, dbn, dw, db, eps=BN_EPS, bn_sums=None):
    """dbn (optional) = dpred (x) w; dw, db +=; bn_sums (optional) += BN-backward channel sums of that gradient"""
    lib = _L()
    C = int(x.shape[-1])
    nvox = x.numel() // C
    _lib.check(_sym('synthsr_head_bwd_ex', x)(_lib.ptr(dpred), _lib.ptr(x), nvox, C, _lib.ptr(stats), _lib.ptr(gamma),
                                       _lib.ptr(beta), eps, _lib.ptr(w), _lib.ptr(dbn), _lib.ptr(dw), _lib.ptr(db),
                                       _lib.ptr(bn_sums), _lib.stream()), 'head_bwd')
    return dbn


def bn_elu_bwd_head(dpred, whead, y, stats, gamma, sums, dbias=None, out=None, eps=BN_EPS):
    """bn_elu_bwd for the BN in front of the head: incoming gradient dpred[v]*whead[c] formed on the fly"""
    lib = _L()
    C = int(y.shape[-1])
    if out is None:
        out = torch.empty_like(y)
    _lib.check(_sym('synthsr_bn_elu_bwd_head', y)(_lib.ptr(dpred), _lib.ptr(whead), _lib.ptr(y), _lib.ptr(out),
                                           _lib.ptr(dbias), y.numel() // C, C, _lib.ptr(stats), _lib.ptr(gamma), eps,
                                           _lib.ptr(sums), _lib.stream()), 'bn_elu_bwd_head')
    return out


def seg_head_fwd(x, stats, gamma, beta, w, b, probs, eps=BN_EPS):
    """probs [nvox, N] = softmax(bn(x) @ w + b): head of the frozen segmentation U-Net"""
    lib = _L()
    C = int(x.shape[-1])
    _lib.check(lib.synthsr_seg_head_fwd(_lib.ptr(x), x.numel() // C, C, _lib.ptr(stats), _lib.ptr(gamma), _lib.ptr(beta),
                                        eps, _lib.ptr(w), _lib.ptr(b), int(probs.shape[-1]), _lib.ptr(probs),
                                        _lib.stream()), 'seg_head_fwd')
    return probs


def seg_dice_sums(probs, seg, cls_idx, cls_gt, sums):
    """sums [2K] (zeroed here) <- soft-Dice numerators / denominators of the K merged classes"""
    lib = _L()
    sums.zero_()
    _lib.check(lib.synthsr_seg_dice_sums(_lib.ptr(probs), _lib.ptr(seg), int(probs.shape[0]), int(probs.shape[1]),
                                         _lib.ptr(cls_idx), _lib.ptr(cls_gt), int(cls_gt.numel()), _lib.ptr(sums),
                                         _lib.stream()), 'seg_dice_sums')
    return sums


def seg_dice_bwd(probs, seg, w, cls_idx, cls_gt, sums, scale, dbn):
    """dbn [nvox, C] = d(scale * dice_loss)/d(BatchNorm output in front of the segmentation head)"""
    lib = _L()
    _lib.check(lib.synthsr_seg_dice_bwd(_lib.ptr(probs), _lib.ptr(seg), int(probs.shape[0]), int(dbn.shape[-1]),
                                        int(probs.shape[1]), _lib.ptr(w), _lib.ptr(cls_idx), _lib.ptr(cls_gt),
                                        int(cls_gt.numel()), _lib.ptr(sums), float(scale), _lib.ptr(dbn), _lib.stream()),
               'seg_dice_bwd')
    return dbn


def adam_step(p, g, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
    lib = _L()
    _lib.check(lib.synthsr_adam_step(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), p.numel(), float(lr_t),
                                     float(beta1), float(beta2), float(eps), float(grad_scale), _lib.stream()),
               'adam_step')


# ---------------------------------------------------------------------- WGAN-GP critic pieces (csrc/critic.hip)
def leaky_relu(x, alpha=0.2, out=None):
    """LeakyReLU (in place when out is None)"""
    out = x if out is None else out
    _lib.check(_L().synthsr_leaky_relu(_lib.ptr(x), None, _lib.ptr(out), x.numel(), float(alpha), _lib.stream()), 'leaky_relu')
    return out


def leaky_relu_bwd(dy, y, alpha=0.2, out=None):
    """dx = dy * (y > 0 ? 1 : alpha); y = the LeakyReLU output"""
    out = torch.empty_like(dy) if out is None else out
    _lib.check(_L().synthsr_leaky_relu(_lib.ptr(y), _lib.ptr(dy), _lib.ptr(out), y.numel(), float(alpha), _lib.stream()),
               'leaky_relu_bwd')
    return out


def dense_fwd(x, W, b, out=None):
    """y [n_out] = b + x [n_in] . W [n_in, n_out]"""
    n_in, n_out = int(W.shape[0]), int(W.shape[1])
    out = torch.empty(n_out, dtype=torch.float32, device=x.device) if out is None else out
    _lib.check(_L().synthsr_dense_fwd(_lib.ptr(x), _lib.ptr(W), _lib.ptr(b), _lib.ptr(out), n_in, n_out, _lib.stream()),
               'dense_fwd')
    return out


def dense_bwd(x, W, dy, dx=None, dW=None):
    """dx [n_in] = W dy (written, if given); dW += x (x) dy (if given)"""
    n_in, n_out = int(W.shape[0]), int(W.shape[1])
    _lib.check(_L().synthsr_dense_bwd(_lib.ptr(x), _lib.ptr(W), _lib.ptr(dy), _lib.ptr(dx), _lib.ptr(dW), n_in, n_out,
                                      _lib.stream()), 'dense_bwd')
    return dx


def axpby(x, y, a, b, out=None):
    out = torch.empty_like(x) if out is None else out
    _lib.check(_L().synthsr_axpby(_lib.ptr(x), _lib.ptr(y), _lib.ptr(out), x.numel(), float(a), float(b), _lib.stream()),
               'axpby')
    return out


def sumsq(x, out):
    """out[0] += sum x^2"""
    _lib.check(_L().synthsr_sumsq(_lib.ptr(x), x.numel(), _lib.ptr(out), _lib.stream()), 'sumsq')
    return out


def bias_leaky_relu(x, bias, alpha=0.2, out=None):
    out = x if out is None else out
    C = int(x.shape[-1])
    _lib.check(_L().synthsr_bias_leaky_relu(_lib.ptr(x), _lib.ptr(bias), _lib.ptr(out), x.numel(), C, float(alpha),
                                            _lib.stream()), 'bias_leaky_relu')
    return out


def colsum(x, out):
    """out [C] += column sums of x [..., C]"""
    C = int(x.shape[-1])
    _lib.check(_L().synthsr_colsum(_lib.ptr(x), x.numel() // C, C, _lib.ptr(out), _lib.stream()), 'colsum')
    return out


# stride-2 'same' Conv3D (even sizes) on the parity kernels of the folded decoder conv; weights [3,3,3,Cin,Cout]
def pack_stride2_weights(w, lo_shape, mode=0, out=None):
    """mode 0: forward weights, mode 1: data-gradient weights (lo_shape = OUTPUT spatial shape)"""
    return pack_conv_weights_ex(w, lo_shape, 0, int(w.shape[3]), mode, up=2, out=out)


def conv3d_stride2(x, wpacked8, Cout, out=None):
    """x [2a,2b,2c,Cin] -> [a,b,c,Cout] (no bias / activation): y[o] = sum_t w[t] x[2o + t]"""
    lib = _L()
    s = x.shape
    lo_shape = (s[0] // 2, s[1] // 2, s[2] // 2)
    if out is None:
        out = torch.empty(lo_shape + (Cout,), dtype=torch.float32, device=x.device)
    _lib.check(lib.synthsr_conv3d_up_dgrad(conv_ctx(), _lib.ptr(x), _lib.ptr(wpacked8), _lib.ptr(out), _lib.i3(lo_shape), int(Cout),
                                           int(s[3]), _lib.stream()), 'conv3d_stride2')
    return out


def conv3d_stride2_dgrad(dy, wpacked8d, Cin, out=None):
    """dy [a,b,c,Cout] -> gradient w.r.t. the stride-2 conv's input [2a,2b,2c,Cin]"""
    lib = _L()
    s = dy.shape
    if out is None:
        out = torch.empty((2 * s[0], 2 * s[1], 2 * s[2], Cin), dtype=torch.float32, device=dy.device)
    _lib.check(lib.synthsr_conv3d_up_fwd(conv_ctx(), _lib.ptr(dy), _lib.ptr(wpacked8d), None, None, _lib.ptr(out), _lib.i3(s[:3]),
                                         int(s[3]), int(Cin), 0, _lib.stream()), 'conv3d_stride2_dgrad')
    return out


def conv3d_stride2_wgrad(x, dy, dw, dwc, dbias=None):
    """dw [3,3,3,Cin,Cout] += sum_o x[2o + t] (x) dy[o]; dwc: scratch [8,27,Cout,Cin] (zeroed here); dbias += sum dy"""
    lib = _L()
    Ci, Co = int(x.shape[3]), int(dy.shape[3])
    dwc.zero_()
    _check_wgrad(lambda: lib.synthsr_conv3d_up_wgrad(conv_ctx(), _lib.ptr(dy), _lib.ptr(x), _lib.ptr(dwc), _lib.i3(dy.shape[:3]), Co, Ci,
                                           _lib.stream()), 'conv3d_stride2_wgrad')
    _lib.check(lib.synthsr_conv3d_stride_unpack(_lib.ptr(dwc), _lib.ptr(dw), Ci, Co, _lib.stream()), 'stride_unpack')
    if dbias is not None:
        colsum(dy, dbias)
    return dw


def mul(x, y, out=None):
    out = torch.empty_like(x) if out is None else out
    _lib.check(_L().synthsr_mul(_lib.ptr(x), _lib.ptr(y), _lib.ptr(out), x.numel(), _lib.stream()), 'mul')
    return out


def lut_gather(labels, lut, out=None):
    """out = lut[labels] (float32; ConvertLabels of the reference): labels int32, lut float32 device tensors"""
    out = torch.empty(labels.shape, dtype=torch.float32, device=labels.device) if out is None else out
    _lib.check(_L().synthsr_lut_gather(_lib.ptr(labels), _lib.ptr(lut), lut.numel(), _lib.ptr(out), labels.numel(),
                                       _lib.stream()), 'lut_gather')
    return out


# ---------------------------------------------------------------------- bf16 convolutions (csrc/conv_bf16.hip)
_bf16_scratch = {}


def pack_conv_weights_bf16(w, mode=0, ci_off=0, cin=None, out=None, up=False):
    """fp32 Keras kernel w [3,3,3,Cin_total,Cout] -> bf16 MFMA fragments (mode 0 forward, 1 data gradient); up: the 8 parity
    sets of the nearest-upsample folding, back to back (conv3d_up / conv3d_up_dgrad)"""
    lib = _L()
    cin_total, cout = int(w.shape[3]), int(w.shape[4])
    cin = cin_total if cin is None else int(cin)
    parities = list(range(8)) if up else [-1]
    n = lib.synthsr_conv3d_bf16_pack_ex(None, None, cin_total, int(ci_off), cin, cout, int(mode), parities[0], None)
    if n < 0:
        _lib.check(int(n), 'conv3d_bf16_pack(size)')
    if out is None:
        out = torch.empty(n * len(parities), dtype=torch.bfloat16, device=w.device)
    assert out.numel() == n * len(parities) and out.dtype == torch.bfloat16
    for k, par in enumerate(parities):
        r = lib.synthsr_conv3d_bf16_pack_ex(_lib.ptr(w), _lib.ptr(out[k * n:(k + 1) * n]), cin_total, int(ci_off), cin, cout,
                                            int(mode), par, _lib.stream())
        if r < 0:
            _lib.check(int(r), 'conv3d_bf16_pack')
    return out


def conv3d_bf16(x, wpacked, bias, Cout, act=1, below=None, stats=None, out=None, alpha=0.0):
    """act(conv3(x) + bias) on bf16 NDHWC tensors (fp32 accumulation); act 2: times ELU'(below); act 3: LeakyReLU(alpha);
    act 4: times LeakyReLU'(below); stats: fp32 [2*Cout] receives the batch mean | variance of the output"""
    lib = _L()
    s = x.shape
    assert x.dtype == torch.bfloat16 and wpacked.dtype == torch.bfloat16
    if out is None:
        out = torch.empty((s[0], s[1], s[2], Cout), dtype=torch.bfloat16, device=x.device)
    nscr = int(lib.synthsr_conv3d_bf16_stats_scratch(_lib.i3(s[:3]), int(s[3]), int(Cout)))
    scratch = _bf16_scratch.get(x.device)   # statistics partials / split-K partial sums (stream-ordered reuse)
    if scratch is None or scratch.numel() < nscr:
        scratch = _bf16_scratch[x.device] = torch.empty(max(nscr, 1 << 20), dtype=torch.float32, device=x.device)
    nscr = scratch.numel()
    with _Timed('conv3d_bf16', s[:3], s[3], Cout):
        _lib.check(lib.synthsr_conv3d_bf16_fwd_ex(_lib.ptr(x), _lib.ptr(wpacked), _lib.ptr(bias), _lib.ptr(out),
                                                  _lib.i3(s[:3]), int(s[3]), int(Cout), int(act), float(alpha),
                                                  _lib.ptr(below), _lib.ptr(stats), _lib.ptr(scratch), nscr,
                                                  _lib.stream()), 'conv3d_bf16_fwd')
    return out


def subsample_odd_bf16(x, below=None, alpha=0.0, out=None):
    """out[o] = x[2 o + 1] (* LeakyReLU'(below[o])): a stride-2 'same' conv = the stride-1 conv at the odd positions"""
    s = x.shape
    lo = (s[0] // 2, s[1] // 2, s[2] // 2)
    if out is None:
        out = torch.empty(lo + (s[3],), dtype=torch.bfloat16, device=x.device)
    _lib.check(_L().synthsr_bf16_subsample_odd(_lib.ptr(x), _lib.ptr(out), _lib.ptr(below), _lib.i3(lo), int(s[3]),
                                               float(alpha), _lib.stream()), 'bf16_subsample_odd')
    return out


def zero_insert_odd_bf16(x, out=None):
    """transpose of subsample_odd_bf16: [2 d0, 2 d1, 2 d2, C] with x at the odd positions, zero elsewhere"""
    s = x.shape
    if out is None:
        out = torch.empty((2 * s[0], 2 * s[1], 2 * s[2], s[3]), dtype=torch.bfloat16, device=x.device)
    _lib.check(_L().synthsr_bf16_zero_insert_odd(_lib.ptr(x), _lib.ptr(out), _lib.i3(s[:3]), int(s[3]), _lib.stream()),
               'bf16_zero_insert_odd')
    return out


def conv3d_wgrad_bf16(x, dz, dw, dbias=None):
    """dw [3,3,3,Cin,Cout] fp32 += weight gradient of bf16 x / dz (dbias += column sums of dz)"""
    lib = _L()
    s = x.shape
    assert x.dtype == torch.bfloat16 and dz.dtype == torch.bfloat16 and dw.dtype == torch.float32
    with _Timed('conv3d_bf16_wgrad', s[:3], s[3], dz.shape[3]):
        _check_wgrad(lambda: lib.synthsr_conv3d_bf16_wgrad(_lib.ptr(x), _lib.ptr(dz), _lib.ptr(dw), _lib.ptr(dbias), _lib.i3(s[:3]),
                                                 int(dw.shape[3]), 0, int(s[3]), int(dz.shape[3]), _lib.stream()),
                   'conv3d_bf16_wgrad')
    return dw


def to_bf16_pad(x, Cd, out=None):
    """fp32 [..., Cs] -> bf16 [..., Cd] with zero-filled extra channels"""
    lib = _L()
    Cs = int(x.shape[-1])
    n = x.numel() // Cs
    if out is None:
        out = torch.empty(tuple(x.shape[:-1]) + (Cd,), dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.synthsr_f32_to_bf16_pad(_lib.ptr(x), _lib.ptr(out), n, Cs, int(Cd), _lib.stream()), 'f32_to_bf16_pad')
    return out
