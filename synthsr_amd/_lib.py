"""ctypes binding of libsynthsr_hip.so (C ABI declared in include/synthsr_hip.h).

There is NO CPU fallback: if the shared library is missing or a call fails, this raises.
torch is used only as the owner of device memory and streams (`tensor.data_ptr()`,
`torch.cuda.current_stream().cuda_stream`).
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# SYNTHSR_HIP_LIB: another build of the same library (A/B timing of kernel variants, tools/); never a different backend
LIB_PATH = os.environ.get('SYNTHSR_HIP_LIB') or os.path.join(_HERE, 'libsynthsr_hip.so')

_lib = None

I3 = c_int * 3
F12 = c_float * 12


class DeformParams(Structure):
    """mirror of synthsr_deform_params"""
    _fields_ = [('in_shape', c_int * 3), ('out_shape', c_int * 3), ('crop', c_int * 3), ('flip', c_int),
                ('has_field', c_int), ('has_affine', c_int), ('half_shape', c_int * 3), ('aff', c_float * 12),
                ('n_channels', c_int), ('lut_size', c_int), ('swap_lut_size', c_int), ('bias_on', c_int * 4),
                ('bias_shape', (c_int * 3) * 4), ('clip_hi', c_float), ('use_philox', c_int),
                ('philox_key', c_uint32 * 2), ('philox_offset', c_uint64), ('chan_first', c_int), ('n_channels_total', c_int),
                ('label_bytes', c_int)]


class ConvCtx(Structure):
    """mirror of synthsr_conv_ctx (include/synthsr_hip.h): the caller-owned context of the fp32 convolutions"""
    _fields_ = [('arithmetic', c_int), ('reserved0', c_int), ('workspace', c_void_p), ('workspace_bytes', c_uint64),
                ('reserved', c_int * 2)]


_P = c_void_p
_S = c_void_p  # stream
_C = POINTER(ConvCtx)  # conv context (None = the library's default: split arithmetic)

SIGNATURES = {
    'synthsr_abi_version': (c_int, []),
    'synthsr_build_arch': (c_char_p, []),
    'synthsr_resize_f32': (c_int, [_P, _P, c_int, POINTER(c_int), POINTER(c_int), c_int, _S]),
    'synthsr_svf_integrate': (c_int, [_P, _P, POINTER(c_int), c_int, _S]),
    'synthsr_affine_resample_linear': (c_int, [_P, _P, c_int, POINTER(c_int), POINTER(c_float), _S]),
    'synthsr_deform_gmm': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, POINTER(DeformParams), _S]),
    'synthsr_mimic_acquisition': (c_int, [_P, _P, POINTER(c_int), POINTER(c_int), POINTER(c_float), POINTER(c_float),
                                          POINTER(c_float), c_int, c_int, c_int, _S]),
    'synthsr_deform_gmm_real': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, POINTER(DeformParams), _S]),
    'synthsr_minmax_init': (c_int, [_P, c_int, _S]),
    'synthsr_minmax_reduce': (c_int, [_P, c_int64, _P, _S]),
    'synthsr_normalise_gamma': (c_int, [_P, _P, c_int64, _P, c_float, _S]),
    'synthsr_blur3d': (c_int, [_P, _P, POINTER(c_int), _P, POINTER(c_int), c_int, c_int, c_int, c_float, _S]),
    'synthsr_normalise_blur2': (c_int, [_P, POINTER(c_int), _P, c_float, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _S]),
    'synthsr_outer3': (c_int, [_P, _P, POINTER(c_int), c_int, c_int, _S]),
    'synthsr_copy_strided': (c_int, [_P, _P, c_int64, c_int, c_int, c_int, c_int, _S]),
    'synthsr_conv3d_pack': (c_int64, [_C, _P, _P, POINTER(c_int), c_int, c_int, c_int, _S]),
    'synthsr_conv3d_fwd': (c_int, [_C, _P, _P, _P, _P, POINTER(c_int), c_int, c_int, c_int, _S]),
    'synthsr_conv3d_fwd_stats': (c_int, [_C, _P, _P, _P, _P, POINTER(c_int), c_int, c_int, c_int, _P, _P, _S]),
    'synthsr_bn_stats_from_partials': (c_int, [_P, c_int, c_int64, c_int, _P, _S]),
    'synthsr_conv3d_fwd_add': (c_int, [_C, _P, _P, _P, _P, _P, POINTER(c_int), c_int, c_int, c_int, _S]),
    'synthsr_split_tile_schedule': (c_int, [c_int, c_int, c_int, c_int, c_int, POINTER(c_int)]),
    'synthsr_conv3d_wgrad_runs_split': (c_int, [_C, POINTER(c_int), c_int, c_int]),
    'synthsr_set_deterministic': (c_int, [c_int]),
    'synthsr_set_deterministic_workspace': (c_int, [_P, c_uint64]),
    'synthsr_deterministic_workspace_demand': (c_uint64, []),
    'synthsr_conv_workspace_bytes': (c_uint64, []),
    'synthsr_deterministic_status': (c_int, []),
    'synthsr_conv3d_plan': (c_int, [_C, POINTER(c_int), c_int, c_int, c_int, POINTER(c_int64)]),
    'synthsr_conv3d_pack_all': (c_int, [_P, _P, _P, c_int, _S]),
    'synthsr_conv3d_pack_ex': (c_int64, [_C, _P, _P, POINTER(c_int), c_int, c_int, c_int, c_int, c_int, c_int, _S]),
    'synthsr_conv3d_up_fwd': (c_int, [_C, _P, _P, _P, _P, _P, POINTER(c_int), c_int, c_int, c_int, _S]),
    'synthsr_conv3d_up_dgrad': (c_int, [_C, _P, _P, _P, POINTER(c_int), c_int, c_int, _S]),
    'synthsr_conv3d_up_wgrad': (c_int, [_C, _P, _P, _P, POINTER(c_int), c_int, c_int, _S]),
    'synthsr_conv3d_up_wgrad_runs_split': (c_int, [_C, POINTER(c_int), c_int, c_int]),
    'synthsr_conv3d_up_unpack': (c_int, [_P, _P, c_int, c_int, c_int, c_int, _S]),
    'synthsr_conv3d_wgrad_bias': (c_int, [_C, _P, _P, _P, _P, POINTER(c_int), c_int, c_int, c_int, c_int, _S]),
    'synthsr_conv3d_wgrad_ex': (c_int, [_C, _P, _P, _P, POINTER(c_int), c_int, c_int, c_int, c_int, _S]),
    'synthsr_conv3d_wgrad': (c_int, [_C, _P, _P, _P, POINTER(c_int), c_int, c_int, _S]),
    'synthsr_elu_bwd_bf16': (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _S]),
    'synthsr_bn_elu_bwd_bf16': (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P, _P, c_float, _P, _S]),
    'synthsr_bn_elu_bwd_head_bf16': (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P, _P, c_float, _P, _S]),
    'synthsr_bn_stats_bf16': (c_int, [_P, c_int64, c_int, _P, _P, _S]),
    'synthsr_bn_maxpool_bf16': (c_int, [_P, _P, POINTER(c_int), c_int, _P, _P, _P, c_float, _S]),
    'synthsr_bn_maxpool_bwd_bf16': (c_int, [_P, _P, _P, POINTER(c_int), c_int, _P, _P, _P, c_float, _S]),
    'synthsr_bn_maxpool_bwd_ex_bf16': (c_int, [_P, _P, _P, POINTER(c_int), c_int, _P, _P, _P, c_float, _P, _S]),
    'synthsr_bn_bwd_reduce_bf16': (c_int, [_P, _P, c_int64, c_int, _P, c_float, _P, _S]),
    'synthsr_upsample_concat_bf16': (c_int, [_P, _P, _P, POINTER(c_int), c_int, c_int, _P, _P, _P, c_float, _S]),
    'synthsr_upsample_concat_bwd_bf16': (c_int, [_P, _P, _P, POINTER(c_int), c_int, c_int, _S]),
    'synthsr_head_loss_fwd_ab': (c_int, [_P, POINTER(c_int), c_int, _P, _P, _P, c_float, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, c_int, POINTER(c_int), _P, _S]),
    'synthsr_head_loss_fwd_ab_bf16': (c_int, [_P, POINTER(c_int), c_int, _P, _P, _P, c_float, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, c_int, POINTER(c_int), _P, _S]),
    'synthsr_head_bwd_from_sums': (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, _S]),
    'synthsr_head_loss_fwd_bf16': (c_int, [_P, POINTER(c_int), c_int, _P, _P, _P, c_float, _P, _P, c_int, _P, c_int, POINTER(c_int), _P, _P, _P, _P, c_int, POINTER(c_int), _S]),
    'synthsr_head_bwd_multi_bf16': (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P, _P, c_float, _P, _P, _P, _P, _S]),
    'synthsr_head_bwd_ex_bf16': (c_int, [_P, _P, c_int64, c_int, _P, _P, _P, c_float, _P, _P, _P, _P, _P, _S]),
    'synthsr_head_bwd_bf16': (c_int, [_P, _P, c_int64, c_int, _P, _P, _P, c_float, _P, _P, _P, _P, _S]),
    'synthsr_conv3d_bf16_pack': (c_int64, [_P, _P, c_int, c_int, c_int, c_int, c_int, _S]),
    'synthsr_conv3d_bf16_pack_ex': (c_int64, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _S]),
    'synthsr_conv3d_bf16_pack_job': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'synthsr_conv3d_bf16_wgrad_part': (c_int, [_P, _P, _P, _P, POINTER(c_int), c_int, c_int, c_int, c_int, _S]),
    'synthsr_conv3d_bf16_up_fwd': (c_int, [_P, _P, _P, POINTER(c_int), c_int, c_int, _S]),
    'synthsr_conv3d_bf16_up_dgrad': (c_int, [_P, _P, _P, POINTER(c_int), c_int, c_int, _P, c_int64, _S]),
    'synthsr_conv3d_bf16_up_wgrad': (c_int, [_P, _P, _P, POINTER(c_int), c_int, c_int, _S]),
    'synthsr_conv3d_bf16_pack_all': (c_int, [_P, _P, _P, c_int, _S]),
    'synthsr_conv3d_bf16_fwd': (c_int, [_P, _P, _P, _P, POINTER(c_int), c_int, c_int, c_int, _P, _P, _P, c_int64, _S]),
    'synthsr_conv3d_bf16_fwd_ex': (c_int, [_P, _P, _P, _P, POINTER(c_int), c_int, c_int, c_int, c_float, _P, _P, _P, c_int64, _S]),
    'synthsr_bf16_subsample_odd': (c_int, [_P, _P, _P, POINTER(c_int), c_int, c_float, _S]),
    'synthsr_bf16_zero_insert_odd': (c_int, [_P, _P, POINTER(c_int), c_int, _S]),
    'synthsr_conv3d_bf16_stats_scratch': (c_int64, [POINTER(c_int), c_int, c_int]),
    'synthsr_conv3d_bf16_wgrad': (c_int, [_P, _P, _P, _P, POINTER(c_int), c_int, c_int, c_int, c_int, _S]),
    'synthsr_f32_to_bf16_pad': (c_int, [_P, _P, c_int64, c_int, c_int, _S]),
    'synthsr_elu_bwd': (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _S]),
    'synthsr_scale_channels': (c_int, [_P, _P, c_int64, c_int, _P, c_int64, _S]),
    'synthsr_elu_bwd_drop': (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P, _P, c_float, _P, _P, _P, _P, c_int64, _S]),
    'synthsr_scale_channels_bf16': (c_int, [_P, _P, c_int64, c_int, _P, c_int64, _S]),
    'synthsr_elu_bwd_drop_bf16': (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P, _P, c_float, _P, _P, _P, _P, c_int64, _S]),
    'synthsr_bn_elu_bwd': (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P, _P, c_float, _P, _S]),
    'synthsr_bn_stats': (c_int, [_P, c_int64, c_int, _P, _P, _S]),
    'synthsr_bn_apply': (c_int, [_P, _P, c_int64, c_int, _P, _P, _P, c_float, _S]),
    'synthsr_bn_apply_bf16': (c_int, [_P, _P, c_int64, c_int, _P, _P, _P, c_float, _S]),
    'synthsr_bn_maxpool': (c_int, [_P, _P, POINTER(c_int), c_int, _P, _P, _P, c_float, _S]),
    'synthsr_bn_maxpool_bwd': (c_int, [_P, _P, _P, POINTER(c_int), c_int, _P, _P, _P, c_float, _S]),
    'synthsr_bn_maxpool_bwd_ex': (c_int, [_P, _P, _P, POINTER(c_int), c_int, _P, _P, _P, c_float, _P, _S]),
    'synthsr_bn_pool_elu_bwd': (c_int, [_P, _P, _P, _P, _P, POINTER(c_int), c_int, _P, _P, _P, _P, c_float, _S]),
    'synthsr_bn_pool_elu_bwd_bf16': (c_int, [_P, _P, _P, _P, _P, POINTER(c_int), c_int, _P, _P, _P, _P, c_float, _S]),
    'synthsr_bn_bwd_reduce': (c_int, [_P, _P, c_int64, c_int, _P, c_float, _P, _S]),
    'synthsr_bn_bwd_apply': (c_int, [_P, _P, _P, c_int64, c_int, _P, _P, c_float, _P, _S]),
    'synthsr_upsample_concat': (c_int, [_P, _P, _P, POINTER(c_int), c_int, c_int, _P, _P, _P, c_float, _S]),
    'synthsr_upsample_concat_bwd': (c_int, [_P, _P, _P, POINTER(c_int), c_int, c_int, _S]),
    'synthsr_head_l1_fwd': (c_int, [_P, c_int64, c_int, _P, _P, _P, c_float, _P, _P, _P, c_int, c_int, _P, _P, _P,
                                    _P, _S]),
    'synthsr_head_loss_fwd': (c_int, [_P, _P, c_int, _P, _P, _P, c_float, _P, _P, c_int, _P, c_int, _P, _P, _P, _P,
                                      _P, c_int, _P, _S]),
    'synthsr_head_bwd_multi': (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P, _P, c_float, _P, _P, _P, _P, _S]),
    'synthsr_ssim_products': (c_int, [_P, _P, _P, _P, _P, _S]),
    'synthsr_ssim_filter': (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, _S]),
    'synthsr_ssim_point': (c_int, [_P, c_int64, c_float, c_float, _P, _P, _S]),
    'synthsr_ssim_combine': (c_int, [_P, _P, _P, _P, _P, _P, _S]),
    'synthsr_conv3d_stride_unpack': (c_int, [_P, _P, c_int, c_int, _S]),
    'synthsr_bias_leaky_relu': (c_int, [_P, _P, _P, c_int64, c_int, c_float, _S]),
    'synthsr_colsum': (c_int, [_P, c_int64, c_int, _P, _S]),
    'synthsr_mul': (c_int, [_P, _P, _P, c_int64, _S]),
    'synthsr_lut_gather': (c_int, [_P, _P, c_int, _P, c_int64, _S]),
    'synthsr_leaky_relu': (c_int, [_P, _P, _P, c_int64, c_float, _S]),
    'synthsr_dense_fwd': (c_int, [_P, _P, _P, _P, c_int64, c_int, _S]),
    'synthsr_dense_bwd': (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _S]),
    'synthsr_axpby': (c_int, [_P, _P, _P, c_int64, c_float, c_float, _S]),
    'synthsr_sumsq': (c_int, [_P, c_int64, _P, _S]),
    'synthsr_bn_elu_bwd_head': (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P, _P, c_float, _P, _S]),
    'synthsr_head_bwd_ex': (c_int, [_P, _P, c_int64, c_int, _P, _P, _P, c_float, _P, _P, _P, _P, _P, _S]),
    'synthsr_seg_head_fwd': (c_int, [_P, c_int64, c_int, _P, _P, _P, c_float, _P, _P, c_int, _P, _S]),
    'synthsr_seg_dice_sums': (c_int, [_P, _P, c_int64, c_int, _P, _P, c_int, _P, _S]),
    'synthsr_seg_dice_bwd': (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P, _P, c_int, _P, c_float, _P, _S]),
    'synthsr_head_bwd': (c_int, [_P, _P, c_int64, c_int, _P, _P, _P, c_float, _P, _P, _P, _P, _S]),
    'synthsr_adam_step': (c_int, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, _S]),
}


CONV_ARITHMETICS = ('fp32_mfma', 'split', 'split9')   # include/synthsr_hip.h: SYNTHSR_ARITH_* (index = value of synthsr_conv_ctx.arithmetic)


class SynthSRHipError(RuntimeError):
    pass


def load():
    """load the shared library (once); raises if it has not been built"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SynthSRHipError('%s not found: build it with `python -m synthsr_amd.build` (hipcc, gfx950). '
                              'There is no CPU fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=''):
    if rc == 0:
        return
    if rc == -1:
        raise ValueError('synthsr_hip: invalid argument / unsupported shape in %s' % what)
    if rc == -3:
        raise SynthSRHipError('synthsr_hip: %s needs scratch and the conv context carries no (or too small a) workspace '
                              '(include/synthsr_hip.h: synthsr_conv_ctx.workspace)' % what)
    raise SynthSRHipError('synthsr_hip: HIP launch error (%d) in %s' % (rc, what))


def i3(seq):
    return I3(*[int(s) for s in seq])


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), 'tensors handed to the HIP library must be contiguous device tensors'
    return c_void_p(t.data_ptr())


def stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
