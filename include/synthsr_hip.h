/*
 * libsynthsr_hip.so — C ABI of the MI355X (gfx950) hot path of SynthSR:
 * on-the-fly synthetic brain generator + 3-D U-Net forward/backward.
 *
 * The reference has no FFI: this path sits behind its Python API / Keras Layer protocol
 * (SURVEY.md §8b).  Each entry point below replaces the TensorFlow graph ops that one
 * reference function emits; the reference file:line it stands in for is cited per function.
 *
 * Conventions
 *   - plain device pointers + explicit sizes; no allocation (scratch is the caller's: synthsr_conv_ctx.workspace),
 *     no ownership transfer, no global
 *     state; every call is stream-ordered on `stream` (a hipStream_t) and re-entrant.  What a
 *     convolution call computes depends on its arguments only: the arithmetic is a field of the
 *     caller-owned synthsr_conv_ctx handed to every conv entry point (NULL = the default), the
 *     launch plans are pure functions of (shape, channels, kind, context).  The one deliberate
 *     exception, the deterministic TEST mode, lives in synthsr_hip_tuning.h and is per device.
 *   - volumes are NDHWC / channels-last, float32 unless stated, 3 spatial dims [d0][d1][d2][C].
 *   - return value: 0 (SYNTHSR_OK) or a negative SYNTHSR_E* code (argument/shape error or a
 *     HIP launch error); the Python layer turns these into the reference's exception types.
 *   - random draws are inputs: either explicit device buffers (parity "tape") or a
 *     (philox key, offset) pair for the in-kernel generator.
 */
#ifndef SYNTHSR_HIP_H
#define SYNTHSR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SYNTHSR_OK 0
#define SYNTHSR_EINVAL (-1)   /* bad argument / unsupported shape */
#define SYNTHSR_ELAUNCH (-2)  /* hip launch error */
#define SYNTHSR_EWORKSPACE (-3) /* the call needs scratch and the context's workspace (or, in the deterministic test mode,
                                 * the registered plane buffer) is missing or too small; nothing was launched */

typedef void* synthsr_stream_t; /* hipStream_t */

/* Context of the fp32 3x3x3 convolutions: caller-owned, read-only during a call, never stored by the library; two contexts
 * (two networks, two threads, two streams of one device) coexist.  Packed weights are only valid under the arithmetic they
 * were packed with -- SPLIT, SPLIT9 and FP32_MFMA each have their own packed layout (SPLIT stacks the three pieces of the
 * Cout = 24 layers, SPLIT9 does not): pack and run under the SAME arithmetic value.
 * workspace: device scratch of workspace_bytes >= synthsr_conv_workspace_bytes() (16 MiB; a bound over every shape) that the
 *   calls made with this context may overwrite: per-workgroup partial sums of BatchNorm statistics and of the first layer's
 *   weight gradient.  Calls sharing a context must be ordered on ONE stream; concurrent streams take one context (one
 *   workspace) each.  A call that needs scratch under a context without one (or NULL) returns SYNTHSR_EWORKSPACE; host-only
 *   queries (plan, pack sizes) and most launches need none.
 *   SYNTHSR_ARITH_SPLIT (default, also what NULL means): every fp32 operand is the exact sum of three bf16 numbers (round to
 *     nearest even on what the previous pieces left); a product a*b is accumulated as a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 +
 *     a2 b0 on v_mfma_f32_16x16x32_bf16, each partial product exact in the fp32 accumulator; what is left out is < 2^-23 |a b|
 *     in the worst case, 2^-27 rms: within the rounding of an fp32 multiply-add.  Inputs, outputs, accumulation, BatchNorm
 *     statistics, gradients and weights stay fp32; against a float64 convolution the result is as accurate as the fp32-MFMA
 *     kernels' (tests/test_split_gpu.py).  Used where the layer has enough 4x4x16 tiles and channel counts that are multiples
 *     of 8 (csrc/conv_split.hip); the rest (first layer, forward / data gradient of the smallest deep layers) runs on the
 *     fp32 matrix instructions in every mode.  The folded decoder convs' weight gradient runs in split arithmetic under this
 *     value only (six products; under SPLIT9 and FP32_MFMA on the fp32 matrix instructions).
 *   SYNTHSR_ARITH_SPLIT9: the same kernels with ALL nine partial products a_i b_j (own packed weights: the three pieces of a
 *     Cout = 24 layer are not stacked): an fp32 product is reproduced exactly at 1.5x the matrix instructions.
 *   SYNTHSR_ARITH_FP32_MFMA: v_mfma_f32_4x4x1 / 16x16x4 kernels everywhere (csrc/conv3d.hip), the round-1/2 path.
 * The reference computes these layers in fp32 on TensorFlow (SynthSR/training.py:330-341). */
#define SYNTHSR_ARITH_FP32_MFMA 0
#define SYNTHSR_ARITH_SPLIT 1
#define SYNTHSR_ARITH_SPLIT9 2
typedef struct synthsr_conv_ctx {
  int arithmetic;           /* SYNTHSR_ARITH_* */
  int reserved0;            /* zero (checked: SYNTHSR_EINVAL otherwise) */
  void* workspace;          /* device scratch owned by the caller, or NULL */
  uint64_t workspace_bytes; /* its size */
  int reserved[2];          /* zero (checked) */
} synthsr_conv_ctx;
/* upper bound of the scratch any call reads or writes in ctx->workspace, whatever the shape */
unsigned long long synthsr_conv_workspace_bytes(void);

/* library / device introspection.  ABI version 2 (round 6): every conv entry point takes the context as its FIRST argument
 * (round 5) and the context carries the caller's workspace; version 1 callers must be rebuilt. */
int synthsr_abi_version(void);
const char* synthsr_build_arch(void);

/* ------------------------------------------------------------------ generator: resampler core */

/* nrn_utils.resize via nrn_layers.Resize (ext/neuron/utils.py:127-154, ext/neuron/layers.py:361-394):
 * sample position i + (i/zoom - i), zoom = out/in; method 0 = linear (edge-clamped trilinear,
 * ext/neuron/utils.py:67-110), 1 = nearest (round-half-even, :113-122). in [i0,i1,i2,C] -> out [o0,o1,o2,C] */
int synthsr_resize_f32(const float* in, float* out, int C, const int in_shape[3], const int out_shape[3],
                       int method, synthsr_stream_t stream);

/* integrate_vec 'ss' branch via VecInt (ext/neuron/utils.py:365-369, ext/neuron/layers.py:241-272):
 * v /= 2^nb_steps; repeat nb_steps: v += transform(v, v).  v, scratch: [s0,s1,s2,3]; result left in v. */
int synthsr_svf_integrate(float* v, float* scratch, const int shape[3], int nb_steps, synthsr_stream_t stream);

/* SpatialTransformer with a single affine (ext/neuron/layers.py:148-151, utils.py:160-219 affine_to_shift,
 * :289-320 transform), linear interpolation.  aff = rows 0..2 of the 4x4 (row-major, 12 floats). */
int synthsr_affine_resample_linear(const float* in, float* out, int C, const int shape[3], const float aff[12],
                                   synthsr_stream_t stream);

/* ------------------------------------------------------------------ generator: fused label->intensity */

typedef struct {
  int in_shape[3];    /* label volume (after optional padding) */
  int out_shape[3];   /* crop shape (RandomCrop, ext/lab2im/layers.py:252-270) */
  int crop[3];        /* crop origin */
  int flip;           /* reverse axis 0 after the crop (RandomFlip, layers.py:391-427) */
  int has_field;      /* elastic part present */
  int has_affine;     /* affine part present (else position = x + u, ext/neuron/layers.py:148-149) */
  int half_shape[3];  /* shape of the integrated half-resolution SVF */
  float aff[12];      /* affine rows 0..2 (identity if no affine) */
  int n_channels;     /* synthetic channels of this launch (<= 4; see chan_first) */
  int lut_size;       /* max(generation_labels)+1 */
  int swap_lut_size;  /* 0 = no L/R swap LUT */
  /* per-channel post-ops fused behind the GMM (BiasFieldCorruption layers.py:1067-1097 and the
   * clip of IntensityAugmentation :1215): */
  int bias_on[4];       /* apply exp(bias) * x */
  int bias_shape[4][3]; /* small bias grid per channel */
  float clip_hi;        /* <=0: no clip */
  int use_philox;       /* 0: noise buffer, 1: in-kernel Philox4x32-10 */
  uint32_t philox_key[2];
  uint64_t philox_offset;
  int chan_first;        /* models with more than 4 synthetic channels run one launch per group of four: first channel of */
  int n_channels_total;  /* this launch (a multiple of 4) and the channel count of the whole model = rows of gmm_lut / of a
                          * noise tape; 0 = this launch carries all channels.  chan_out, minmax, bias_on / bias_shape and
                          * bias_small are the GROUP's; in-kernel noise: the group index enters the Philox counter */
  int label_bytes;    /* element size of `labels`: 0 or 4 = int32 (the reference's label maps), 1 = uint8, 2 = int16 (a label
                       * pool kept on the device in the narrowest type its values fit: model_inputs.py:104-107 loads them as int) */
} synthsr_deform_params;

/* Fuses RandomSpatialDeformation's final SpatialTransformer(nearest) (layers.py:200-203;
 * combine_non_linear_and_aff_to_shift utils.py:222-286; full-res Resize of the field layers.py:196),
 * RandomCrop, RandomFlip (+LUT swap), SampleConditionalGMM (layers.py:480-498), BiasFieldCorruption and
 * the clip + min/max reduction of IntensityAugmentation (layers.py:1214-1231).
 *   labels   int32 [in_shape] (uint8 / int16 with p->label_bytes = 1 / 2)   field_half float [half_shape,3] (may be NULL)
 *   gmm_lut  float [2][n_channels][lut_size]  (means then stds)
 *   swap_lut int32 [swap_lut_size] or NULL    noise float [out_shape, n_channels] or NULL (philox)
 *   bias_small float, concatenated small grids of the channels with bias_on (already scaled by std)
 *   seg_out  int32 [out_shape] or NULL        chan_out float planar [n_channels][out_shape]
 *   minmax   uint32 [n_channels][2] ordered-encoded min/max, must be initialised by synthsr_minmax_init */
int synthsr_deform_gmm(const int32_t* labels, const float* field_half, const float* gmm_lut,
                       const int32_t* swap_lut, const float* noise, const float* bias_small, int32_t* seg_out,
                       float* chan_out, uint32_t* minmax, const synthsr_deform_params* p,
                       synthsr_stream_t stream);

/* the same with a real scan as regression target (output_channel=None, labels_to_image_model.py:109-113,126-160):
 * real_in float [in_shape] is sampled at the SAME positions with inter_method 'linear' (edge-clamped trilinear), cropped
 * and flipped like the labels -> real_out float [out_shape]; its min/max (for IntensityAugmentation(normalise), :250)
 * accumulate into real_minmax[2] (ordered-encoded, initialised by synthsr_minmax_init).  real_in == NULL: plain call. */
int synthsr_deform_gmm_real(const int32_t* labels, const float* field_half, const float* gmm_lut,
                            const int32_t* swap_lut, const float* noise, const float* bias_small, int32_t* seg_out,
                            float* chan_out, uint32_t* minmax, const float* real_in, float* real_out,
                            uint32_t* real_minmax, const synthsr_deform_params* p, synthsr_stream_t stream);

int synthsr_minmax_init(uint32_t* minmax, int n_pairs, synthsr_stream_t stream);
/* generic min/max over n floats into one ordered-encoded pair (IntensityAugmentation on a real image) */
int synthsr_minmax_reduce(const float* x, int64_t n, uint32_t* minmax, synthsr_stream_t stream);

/* IntensityAugmentation normalise + gamma (layers.py:1235-1242): x = ((clip(x,m,M)-m)/(M-m+1e-7))^gexp,
 * m,M decoded on device from `minmax`; in place allowed. gexp = exp(gamma) computed on host; <=0: skip pow */
int synthsr_normalise_gamma(const float* x, float* out, int64_t n, const uint32_t* minmax, float gexp,
                            synthsr_stream_t stream);

/* GaussianBlur (layers.py:732-767): tf.nn.conv3d(...,'SAME') of one channel with a full 3-D kernel built on
 * the host (edit_tensors.py:86-181).  out element o is written at out[o*out_stride + out_offset]; if
 * fill_offset >= 0 the value fill_value is also written at out[o*out_stride + fill_offset]
 * (all-ones reliability map, edit_tensors.py:333). kernel: device float [k0*k1*k2]. */
int synthsr_blur3d(const float* in, float* out, const int shape[3], const float* kernel, const int ksize[3],
                   int out_stride, int out_offset, int fill_offset, float fill_value, synthsr_stream_t stream);

/* One channel of the common case of SynthSR/labels_to_image_model.py:184-228 in a single pass over HBM:
 * IntensityAugmentation's min-max normalisation + gamma (ext/lab2im/layers.py:1227-1242; x = the clipped channel, minmax =
 * its running extremes as left by synthsr_deform_gmm, gexp = exp(.5 z)) -> GaussianBlur(.5) (:186, kernel1 3x3x3) written to
 * `target` (one contiguous channel: the regression-target tap, :189-196) -> the acquisition blur (:223, kernel2 3x3x3) written
 * to image[v * image_stride + image_offset], with image[v * image_stride + fill_offset] = fill_value (the all-ones
 * reliability map of a channel that is not down-sampled, :228) when fill_offset >= 0.  Same arithmetic, tap order and zero
 * padding as synthsr_normalise_gamma followed by two synthsr_blur3d calls (bit-identical results); x must not alias the
 * outputs. */
int synthsr_normalise_blur2(const float* x, const int shape[3], const uint32_t* minmax, float gexp, const float* kernel1,
                            const float* kernel2, float* target, float* image, int image_stride, int image_offset,
                            int fill_offset, float fill_value, synthsr_stream_t stream);

/* MimicAcquisition (ext/lab2im/layers.py:927-990; randomise_res path, labels_to_image_model.py:220) with
 * min_subsample_res = volume_res: nearest down-sampling by down_zoom then linear up-sampling by up_zoom (both as the
 * reference computes them from the sampled resolution, passed in as float32), fused per output voxel.
 * out[o*out_stride + out_offset] = resampled value; if dist_offset >= 0, out[o*out_stride + dist_offset] = distance (mm)
 * of the output voxel to the nearest acquired grid point (the 'distance map' that replaces the reliability map).
 * out_offset < 0: only the distance map is written. */
int synthsr_mimic_acquisition(const float* in, float* out, const int in_shape[3], const int out_shape[3],
                              const float down_zoom[3], const float up_zoom[3], const float subsample_res[3],
                              int out_stride, int out_offset, int dist_offset, synthsr_stream_t stream);

/* reliability map (edit_tensors.py:313-329): out[o*stride+offset] = w0[o0]*w1[o1]*w2[o2]; w concatenated */
int synthsr_outer3(const float* w, float* out, const int shape[3], int out_stride, int out_offset,
                   synthsr_stream_t stream);

/* strided copy: out[i*out_stride+out_offset] = in[i*in_stride+in_offset] */
int synthsr_copy_strided(const float* in, float* out, int64_t n, int in_stride, int in_offset, int out_stride,
                         int out_offset, synthsr_stream_t stream);

/* ------------------------------------------------------------------ U-Net (ext/neuron/models.py:256-498) */

/* Keras Conv3D(k=3,'same') weights are [3][3][3][Cin][Cout].  Packs them into the MFMA B-fragment order
 * used by synthsr_conv3d_fwd for a layer of spatial size `shape` (the output-channel tiling depends on the
 * launch geometry chosen for that size).  mode 0: forward; mode 1: data-gradient (taps flipped, Cin<->Cout
 * swapped; `shape`, Cin, Cout are still those of the FORWARD layer).
 * Returns the number of floats written (or required if packed==NULL), negative on error. */
int64_t synthsr_conv3d_pack(const synthsr_conv_ctx* ctx, const float* w, float* packed, const int shape[3], int Cin, int Cout,
                            int mode, synthsr_stream_t stream);

/* Conv3D 3x3x3 'same' + bias + activation (0 linear, 1 ELU alpha=1) — models.py:316,444.
 * in [d0,d1,d2,Cin], out [d0,d1,d2,Cout]; wpacked from synthsr_conv3d_pack(shape,...) with the same shape.
 * For the data-gradient call it with (dout, pack(mode 1), NULL, din, shape, Cout, Cin, 0). bias may be NULL. */
int synthsr_conv3d_fwd(const synthsr_conv_ctx* ctx, const float* in, const float* wpacked, const float* bias, float* out,
                       const int shape[3], int Cin, int Cout, int act, synthsr_stream_t stream);

/* synthsr_conv3d_fwd followed by synthsr_bn_stats(out) -- BatchNorm batch statistics of the layer output (stats[2C],
 * ws = 2C doubles of scratch).  For the layers that run on the 4x4x1 kernel the sums are accumulated in the conv
 * epilogue (per-workgroup partials + a tiny reduction; no extra pass over the activation); otherwise the two calls are
 * simply chained. */
int synthsr_conv3d_fwd_stats(const synthsr_conv_ctx* ctx, const float* in, const float* wpacked, const float* bias, float* out,
                             const int shape[3], int Cin, int Cout, int act, float* stats, double* ws, synthsr_stream_t stream);

/* act 0/1: out = act(conv3(in) + addend + bias); addend is indexed like out and may alias it (in-place accumulation).
 * Layers that are split over input channels (small deep levels) accumulate with atomics and require addend == out or NULL.
 * act 2 (data gradient fused with the ELU backward of the layer below): out = conv3(in) * elu'(y), y = addend != out is
 * that layer's ELU output, elu'(y) = 1 for y > 0 else y + 1 (layers.py ELU, models.py:283). */
int synthsr_conv3d_fwd_add(const synthsr_conv_ctx* ctx, const float* in, const float* wpacked, const float* bias,
                           const float* addend, float* out, const int shape[3], int Cin, int Cout, int act,
                           synthsr_stream_t stream);

/* --- nearest-upsample folding (decoder conv on concatenate([skip, UpSampling3D(2)(lo)]), models.py:426-444) ------
 * A 3x3x3 conv over an up-sampled tensor equals 8 parity-wise 2x2x2 convs over the low-res tensor with summed taps
 * (3.4x fewer FLOPs, same result up to float32 re-association).  The layer is evaluated as
 *   out = act( conv3(skip; W[:, :Cs]) + upconv(lo; W[:, Cs:]) + bias ).
 * `w` is the layer's Keras kernel [3][3][3][Cin_total][Cout]; (ci_off, Cin) selects the input-channel range. */
int64_t synthsr_conv3d_pack_ex(const synthsr_conv_ctx* ctx, const float* w, float* packed, const int shape[3], int Cin_total,
                               int ci_off, int Cin, int Cout, int mode,
                               int up /* 0 plain, 1: 8 parity weight sets, shape = lo shape */, synthsr_stream_t stream);
/* out[2*lo_shape, Cout] = act(upconv(lo [lo_shape, Cl]) + addend + bias); wpacked8 from pack_ex(..., mode 0, up 1) */
int synthsr_conv3d_up_fwd(const synthsr_conv_ctx* ctx, const float* lo, const float* wpacked8, const float* bias,
                          const float* addend, float* out, const int lo_shape[3], int Cl, int Cout, int act,
                          synthsr_stream_t stream);
/* dlo[lo_shape, Cl] = adjoint of upconv applied to dout [2*lo_shape, Cout]; wpacked8 from pack_ex(..., mode 1, up 1) */
int synthsr_conv3d_up_dgrad(const synthsr_conv_ctx* ctx, const float* dout, const float* wpacked8, float* dlo,
                            const int lo_shape[3], int Cl, int Cout, synthsr_stream_t stream);
/* per-parity weight gradients dwc[8][27][Cl][Cout] (ACCUMULATED: zeros before the first use, see _up_unpack) ...  (takes the context like every conv entry point of ABI 2 -- under split
 * arithmetic with Cl % 16 == 0 and Cout % 24 == 0 it runs in split arithmetic, conv_split.hip conv3d_split_upwgrad_kernel) */
int synthsr_conv3d_up_wgrad(const synthsr_conv_ctx* ctx, const float* lo, const float* dout, float* dwc, const int lo_shape[3],
                            int Cl, int Cout, synthsr_stream_t stream);
/* 1 / 0: whether that launch runs on the split kernel under this context (the dispatcher's own condition) */
int synthsr_conv3d_up_wgrad_runs_split(const synthsr_conv_ctx* ctx, const int lo_shape[3], int Cl, int Cout);
/* ... folded back onto the original taps: dw[27][Cin_total][Cout] (+=) at channels [ci_off, ci_off+Cl).  The partials are
 * CONSUMED: on return (stream-ordered) every slot the weight-gradient kernels can write is zero again, so dwc needs the caller's
 * zeros only before its first use (any prefix of a zeroed buffer may serve a smaller layer later) */
int synthsr_conv3d_up_unpack(float* dwc, float* dw, int Cin_total, int ci_off, int Cl, int Cout, synthsr_stream_t stream);
/* weight gradient of a layer part: in has Cin channels, dw rows are Cin_total wide, written at ci_off */
/* as synthsr_conv3d_wgrad_ex, and dbias[Cout] += sum over voxels of dout (a constant-1 row of the same GEMM); NULL = skip */
int synthsr_conv3d_wgrad_bias(const synthsr_conv_ctx* ctx, const float* in, const float* dout, float* dw, float* dbias,
                              const int shape[3], int Cin_total, int ci_off, int Cin, int Cout, synthsr_stream_t stream);
int synthsr_conv3d_wgrad_ex(const synthsr_conv_ctx* ctx, const float* in, const float* dout, float* dw, const int shape[3],
                            int Cin_total, int ci_off, int Cin, int Cout, synthsr_stream_t stream);

/* launch geometry the kernels will use: out = {chunk width CK, #ci chunks, n-tiles per workgroup (0: 4x4x1-MFMA layout of
 * the Cout = 24 layers, -Cin: first-layer layout), #n chunks, MT, ksplit (split arithmetic: 2 = the 512-thread split-K-halves
 * kernel of the layers that leave CUs single-occupied), NV, floats per packed weight set}.
 * kind: 1 plain conv; 2 forward parity convs of a folded decoder conv; 0 their data gradient */
int synthsr_conv3d_plan(const synthsr_conv_ctx* ctx, const int shape[3], int CinE, int CoutE, int kind, int64_t out[8]);
/* 1 / 0: whether the weight gradient of a plain 3x3x3 conv of this shape runs on the split kernels under this context (the
 * dispatcher's own condition) -- what benchmarks price a layer against.  Negative: SYNTHSR_EINVAL. */
int synthsr_conv3d_wgrad_runs_split(const synthsr_conv_ctx* ctx, const int shape[3], int Cin, int Cout);
/* packs every layer of a network in ONE launch.  jobs_dev: int64 [njobs][14] = {w_off, dst_off, count, cin_total,
 * ci_off, cin, cout, mode, ck, ncc, nt, parity(-1 plain), nv, mfma_count}; w_off / dst_off are float offsets.
 * `packed` must have been ZEROED once by the caller: of a 27-slot parity set (parity 0..7, nt > 0) only the 8 slots of the
 * parity's 2x2x2 window are written, the structurally empty 19 are left as they are */
int synthsr_conv3d_pack_all(const float* params, float* packed, const int64_t* jobs_dev, int njobs,
                            synthsr_stream_t stream);

/* (rounds 1-4 had a process-wide option switch and a process-wide arithmetic setter behind these kernels; both are gone --
 * plan parameters are constants of the library, the arithmetic travels in synthsr_conv_ctx.  What is left in
 * synthsr_hip_tuning.h: the deterministic TEST mode and a host-only tile-schedule query.) */

/* ---- bf16 twins of the HBM-bound U-Net kernels: same arguments and semantics as the float32 entry point of the same
 * name (which cites the reference layers it replaces); activation / activation-gradient tensors are NDHWC bfloat16
 * (void*), parameters, statistics, reductions, predictions and losses stay float32; all arithmetic is float32, stores
 * round to nearest even ("bf16 with fp32 norm accum", BASELINE.json configs[3]) */
int synthsr_elu_bwd_bf16(const void* dy, const void* dy2, const void* y, void* dz, float* dbias, int64_t nvox, int C,
                         synthsr_stream_t stream);
int synthsr_bn_elu_bwd_bf16(const void* dy, const void* dy2, const void* y, void* dz, float* dbias, int64_t nvox, int
                            C, const float* stats, const float* gamma, float eps, const float* sums, synthsr_stream_t
                            stream);
int synthsr_bn_elu_bwd_head_bf16(const float* dpred, const float* whead, const void* y, void* dz, float* dbias,
                                 int64_t nvox, int C, const float* stats, const float* gamma, float eps, const float*
                                 sums, synthsr_stream_t stream);
int synthsr_bn_stats_bf16(const void* x, int64_t nvox, int C, float* stats, double* ws, synthsr_stream_t stream);
int synthsr_bn_maxpool_bf16(const void* x, void* y, const int shape[3], int C, const float* stats, const float* gamma,
                            const float* beta, float eps, synthsr_stream_t stream);
int synthsr_bn_maxpool_bwd_bf16(const void* dy, const void* x, void* dbn, const int shape[3], int C, const float*
                                stats, const float* gamma, const float* beta, float eps, synthsr_stream_t stream);
int synthsr_bn_maxpool_bwd_ex_bf16(const void* dy, const void* x, void* dbn, const int shape[3], int C, const float*
                                   stats, const float* gamma, const float* beta, float eps, float* sums,
                                   synthsr_stream_t stream);
int synthsr_bn_bwd_reduce_bf16(const void* dy, const void* x, int64_t nvox, int C, const float* stats, float eps,
                               float* sums, synthsr_stream_t stream);
int synthsr_upsample_concat_bf16(const void* skip, const void* lo, void* out, const int shape[3], int Cs, int Cl,
                                 const float* stats, const float* gamma, const float* beta, float eps,
                                 synthsr_stream_t stream);
int synthsr_upsample_concat_bwd_bf16(const void* dcat, void* dskip, void* dlo_bn, const int shape[3], int Cs, int Cl,
                                     synthsr_stream_t stream);
int synthsr_head_loss_fwd_bf16(const void* x, const int* shape, int C, const float* stats, const float* gamma, const
                               float* beta, float eps, const float* w, const float* b, int K, const float* residual,
                               int res_stride, const int* res_offs, const float* target, float* pred, float* dpred,
                               float* loss, int kind, const int* crop, synthsr_stream_t stream);
int synthsr_head_bwd_multi_bf16(const float* dpred, const void* x, int64_t nvox, int C, int K, const float* stats,
                                const float* gamma, const float* beta, float eps, const float* w, void* dbn, float*
                                dw, float* db, synthsr_stream_t stream);
int synthsr_head_bwd_ex_bf16(const float* dpred, const void* x, int64_t nvox, int C, const float* stats, const float*
                             gamma, const float* beta, float eps, const float* w, void* dbn, float* dw, float* db,
                             float* bn_sums, synthsr_stream_t stream);
int synthsr_head_bwd_bf16(const float* dpred, const void* x, int64_t nvox, int C, const float* stats, const float*
                          gamma, const float* beta, float eps, const float* w, void* dbn, float* dw, float* db,
                          synthsr_stream_t stream);

/* ---- bf16 variants (BASELINE.json configs[3] / [4]: bf16 activations and weights, fp32 accumulation, fp32 BatchNorm
 * statistics; csrc/conv_bf16.hip).  Activation / gradient tensors are NDHWC bfloat16 (passed as void*), parameters,
 * parameter gradients and statistics stay float32 (fp32 master weights live in the optimizer's flat buffer).  They
 * replace the same Keras layers as the float32 entry points (ext/neuron/models.py:297-299,412-414; the reference has no
 * reduced-precision path: parity is stated against the float32 oracle with a bf16 tolerance).
 * pack: fp32 Keras kernel w [27][Cin_total][Cout] -> bf16 MFMA A-fragments of channels [ci_off, ci_off + Cin); mode 0
 * forward, 1 data gradient; returns the number of bf16 values (packed == NULL: size query).  The tensor the layer reads
 * has ceil(CinE / 8) * 8 channels (pad channels get zero weights); CoutE % 4 == 0 */
int64_t synthsr_conv3d_bf16_pack(const float* w, void* packed, int Cin_total, int ci_off, int Cin, int Cout, int mode,
                                 synthsr_stream_t stream);
/* parity -1: as synthsr_conv3d_bf16_pack.  parity 0..7 (= 4 pz + 2 py + px): one of the 8 weight sets of the
 * nearest-upsample folding of a decoder's first conv (UpSampling3D(2) -> concatenate -> Conv3D, ext/neuron/models.py:
 * 426-444): 8 taps on the LOW-resolution tensor, each the sum of the original taps that fall on it under that output
 * parity (p = 0: low-res offsets {-1, 0} <- taps {0}, {1, 2}; p = 1: offsets {0, +1} <- {0, 1}, {2}); mode 0 = the parity
 * conv itself (synthsr_conv3d_bf16_up_fwd), mode 1 = its data gradient (synthsr_conv3d_bf16_up_dgrad). */
int64_t synthsr_conv3d_bf16_pack_ex(const float* w, void* packed, int Cin_total, int ci_off, int Cin, int Cout, int mode,
                                    int parity, synthsr_stream_t stream);
/* One launch for every packed weight set of a network: jobs_dev[njobs][13] int64 = {w_off (floats into params), dst_off
 * (bf16 elements into packed), then the 11 fields synthsr_conv3d_bf16_pack_job fills in}.  Replaces the per-layer Keras weight
 * reads of the reference's Conv3D layers (ext/neuron/models.py:297-316) after every optimizer step. */
int synthsr_conv3d_bf16_pack_job(int Cin_total, int ci_off, int Cin, int Cout, int mode, int parity, int64_t job[13]);
int synthsr_conv3d_bf16_pack_all(const float* params, void* packed, const int64_t* jobs_dev, int njobs,
                                 synthsr_stream_t stream);

/* out = act(conv3(in) + bias); act 0 linear, 1 ELU, 2 multiply by ELU'(below) (data gradient fused with the ELU backward
 * of the layer below; below = that layer's ELU output [vox][Cout]), 5 ELU(conv3(in) + bias + below) (below = partial sums
 * of another input-channel range, may be `out` itself: the skip-channel half of a folded decoder conv).  stats != NULL: BatchNorm batch statistics
 * [mean Cout | var Cout] of the (bf16-rounded) output, accumulated in fp32 / double.  `scratch` (>=
 * synthsr_conv3d_bf16_stats_scratch floats; may be NULL when no statistics are wanted) also holds the fp32 partial sums
 * of the split-K path that small volumes take */
int synthsr_conv3d_bf16_fwd(const void* in, const void* wp, const float* bias, void* out, const int shape[3], int Cin,
                            int Cout, int act, const void* below, float* stats, float* scratch, int64_t scratch_floats,
                            synthsr_stream_t stream);
int64_t synthsr_conv3d_bf16_stats_scratch(const int shape[3], int Cin, int Cout);
/* same with the LeakyReLU epilogues of the WGAN-GP critic (SynthSR/fine_tuning_with_adversary.py:482-508):
 * act 3 = LeakyReLU(alpha), act 4 = multiply by LeakyReLU'(below) (below = the layer's activation output) */
int synthsr_conv3d_bf16_fwd_ex(const void* in, const void* wp, const float* bias, void* out, const int shape[3], int Cin,
                               int Cout, int act, float alpha, const void* below, float* stats, float* scratch,
                               int64_t scratch_floats, synthsr_stream_t stream);
/* stride-2 'same' Conv3D of an even-sized volume (TensorFlow pads (0, 1)) = the stride-1 conv sampled at the odd positions:
 * out[o] = in[2 o + 1] (optionally times LeakyReLU'(below[o])); zero_insert is its transpose (gradient w.r.t. the
 * full-resolution tensor).  bf16 NDHWC, C % 4 == 0 */
int synthsr_bf16_subsample_odd(const void* in, void* out, const void* below, const int out_shape[3], int C, float alpha,
                               synthsr_stream_t stream);
int synthsr_bf16_zero_insert_odd(const void* in, void* out, const int in_shape[3], int C, synthsr_stream_t stream);
/* dw[27][Cin][Cout] (fp32) += sum_v in[v + t - 1][ci] * dout[v][co];  dbias[Cout] += sum_v dout[v]  (may be NULL);
 * both zeroed by the caller.  `in` has Cin channels (Cin % 8 == 0), dw covers its first Cin_total <= Cin channels;
 * Cout % 8 == 0 */
int synthsr_conv3d_bf16_wgrad(const void* in, const void* dout, float* dw, float* dbias, const int shape[3],
                              int Cin_total, int ci_off, int Cin, int Cout, synthsr_stream_t stream);
/* the same for the input-channel range [ci_off, ci_off + Cin) of a layer with Cin_total input channels (`in` = that
 * range as its own tensor, Cin % 8 == 0): the skip-channel half of a folded decoder conv (SynthSR/../models.py:434:
 * concatenate([skip, up]) is never materialised) */
int synthsr_conv3d_bf16_wgrad_part(const void* in, const void* dout, float* dw, float* dbias, const int shape[3],
                                   int Cin_total, int ci_off, int Cin, int Cout, synthsr_stream_t stream);
/* Nearest-upsample folding in bf16 (the float32 versions: synthsr_conv3d_up_fwd / _up_dgrad / _up_wgrad): the part of a
 * decoder's first conv that reads UpSampling3D(2)(lo) is evaluated on the low-resolution tensor lo [lo_shape][Cl] as 8
 * parity convs with 2x2x2 taps (3.4x fewer FLOPs, no up-sampled / concatenated tensor).
 *  up_fwd:   out [2 lo_shape][Cout] = the raw partial sums (bf16-rounded; bias / activation come with the skip-channel
 *            conv, synthsr_conv3d_bf16_fwd act 5); wpacked8 = the 8 parity sets (pack_ex mode 0), back to back;
 *  up_dgrad: dlo [lo_shape][Cl] = gradient w.r.t. lo of dout [2 lo_shape][Cout]; wpacked8 = pack_ex mode 1 sets; scratch:
 *            fp32 partial planes of the split-K path of small volumes (may be NULL);
 *  up_wgrad: dwc [8][27][Cl][Cout] fp32 (zeros before the first use) += per-parity gradients in 27-slot form;
 *            synthsr_conv3d_up_unpack then folds them onto dw[27][Cin_total][Cout] and leaves dwc zeroed again. */
int synthsr_conv3d_bf16_up_fwd(const void* lo, const void* wpacked8, void* out, const int lo_shape[3], int Cl, int Cout,
                               synthsr_stream_t stream);
int synthsr_conv3d_bf16_up_dgrad(const void* dout, const void* wpacked8, void* dlo, const int lo_shape[3], int Cl, int Cout,
                                 float* scratch, int64_t scratch_floats, synthsr_stream_t stream);
int synthsr_conv3d_bf16_up_wgrad(const void* lo, const void* dout, float* dwc, const int lo_shape[3], int Cl, int Cout,
                                 synthsr_stream_t stream);
/* float32 [n][Cs] -> bfloat16 [n][Cd], Cd >= Cs, zero fill (the generator's image -> first-layer input, Cin 2 -> 8) */
int synthsr_f32_to_bf16_pad(const float* src, void* dst, int64_t n, int Cs, int Cd, synthsr_stream_t stream);

/* weight gradient: dw[3][3][3][Cin][Cout] += sum_v in[v+t-1][ci] * dout[v][co]   (dw must be zeroed by caller) */
int synthsr_conv3d_wgrad(const synthsr_conv_ctx* ctx, const float* in, const float* dout, float* dw, const int shape[3], int Cin,
                         int Cout, synthsr_stream_t stream);

/* dz = dy * ELU'(y) (y = saved activation output), optional dy2 added first (skip-connection gradient),
 * dbias[c] += sum_v dz[v][c]  (dbias zeroed by caller; may be NULL) */
int synthsr_elu_bwd(const float* dy, const float* dy2, const float* y, float* dz, float* dbias, int64_t nvox, int C,
                    synthsr_stream_t stream);

/* fused BN backward + ELU backward: dy is the gradient w.r.t. BN(y); y the conv+ELU output that the BN normalised;
 * sums from synthsr_bn_bwd_reduce.  dz = (bn_bwd(dy) + dy2) * ELU'(y); dbias += sum dz */
int synthsr_bn_elu_bwd(const float* dy, const float* dy2, const float* y, float* dz, float* dbias, int64_t nvox, int C,
                       const float* stats, const float* gamma, float eps, const float* sums, synthsr_stream_t stream);

/* the same for the BN in front of the 1x1x1 head, whose output gradient is rank-1: dy[v][c] = dpred[v]*whead[c] is
 * formed on the fly (never stored); sums from synthsr_head_bwd_ex */
int synthsr_bn_elu_bwd_head(const float* dpred, const float* whead, const float* y, float* dz, float* dbias, int64_t nvox,
                            int C, const float* stats, const float* gamma, float eps, const float* sums,
                            synthsr_stream_t stream);

/* Feature-wise dropout with ONE MASK PER SAMPLE of a batch (KL.Dropout(rate, noise_shape=[None, 1, 1, 1, C]),
 * ext/neuron/models.py:320-324, 448-451; `batchsize > 1` together with `dropout > 0`): the volumes of a batch are stacked
 * along the first axis (nvox_per_sample voxels each), scale [B][C] holds 0 or 1 / (1 - rate).
 *   synthsr_scale_channels: out[v][c] = x[v][c] * scale[sample(v)][c] (the dropped-out tensor, or a gradient; in place allowed)
 *   synthsr_elu_bwd_drop:   the three ELU-backward entry points above for a conv output y whose consumer (the next conv, or the
 *     BatchNorm when stats / gamma / sums are given; dpred / whead: the rank-1 gradient of the head) read d = scale * y: the
 *     incoming gradient is w.r.t. d (BatchNorm: x-hat from d), is multiplied by scale, dy2 (the skip connection reads y itself)
 *     is added unscaled, then * ELU'(y); dbias += sum dz */
int synthsr_scale_channels(const float* x, float* out, int64_t nvox, int C, const float* scale, int64_t nvox_per_sample,
                           synthsr_stream_t stream);
int synthsr_elu_bwd_drop(const float* dy, const float* dy2, const float* y, float* dz, float* dbias, int64_t nvox, int C,
                         const float* stats, const float* gamma, float eps, const float* sums, const float* dpred,
                         const float* whead, const float* drop, int64_t nvox_per_sample, synthsr_stream_t stream);
/* (their bf16 twins: activations / activation gradients bfloat16, scale / statistics / sums / dpred / whead / dbias float32) */
int synthsr_scale_channels_bf16(const void* x, void* out, int64_t nvox, int C, const float* scale, int64_t nvox_per_sample,
                                synthsr_stream_t stream);
int synthsr_elu_bwd_drop_bf16(const void* dy, const void* dy2, const void* y, void* dz, float* dbias, int64_t nvox, int C,
                              const float* stats, const float* gamma, float eps, const float* sums, const float* dpred,
                              const float* whead, const float* drop, int64_t nvox_per_sample, synthsr_stream_t stream);

/* BatchNormalization(axis=-1), training mode (models.py:351,477; Keras 2.3.1 semantics, eps=1e-3):
 * stats[0..C) = mean, stats[C..2C) = biased variance over the nvox voxels */
int synthsr_bn_stats(const float* x, int64_t nvox, int C, float* stats, double* ws /* 2C doubles scratch */,
                     synthsr_stream_t stream);
/* mean | biased variance from per-workgroup partial sums: partial[nwg][2C] = (sum | sum of squares) over nvox voxels in total */
int synthsr_bn_stats_from_partials(const float* partial, int nwg, int64_t nvox, int C, float* stats,
                                   synthsr_stream_t stream);
/* y = gamma*(x-mean)*rsqrt(var+eps)+beta */
int synthsr_bn_apply(const float* x, float* y, int64_t nvox, int C, const float* stats, const float* gamma,
                     const float* beta, float eps, synthsr_stream_t stream);
int synthsr_bn_apply_bf16(const void* x, void* y, int64_t nvox, int C, const float* stats, const float* gamma,
                          const float* beta, float eps, synthsr_stream_t stream);
/* fused BN apply + MaxPooling3D(2) (models.py:356): x [d0,d1,d2,C] -> y [d0/2,d1/2,d2/2,C] */
int synthsr_bn_maxpool(const float* x, float* y, const int shape[3], int C, const float* stats, const float* gamma,
                       const float* beta, float eps, synthsr_stream_t stream);
/* backward of the fused BN + max-pool: dy [d/2], x [d] -> dbn [d] (gradient w.r.t. the BN output) */
int synthsr_bn_maxpool_bwd(const float* dy, const float* x, float* dbn, const int shape[3], int C,
                           const float* stats, const float* gamma, const float* beta, float eps,
                           synthsr_stream_t stream);
/* the same, and sums[2C] += the channel sums synthsr_bn_bwd_reduce(dbn, x) would compute for the BatchNorm below the pool
 * (sum dbn | sum dbn*xhat): the routed gradient is 7/8 zeros and already in registers here.  sums == NULL: plain. */
int synthsr_bn_maxpool_bwd_ex(const float* dy, const float* x, float* dbn, const int shape[3], int C, const float* stats,
                              const float* gamma, const float* beta, float eps, float* sums, synthsr_stream_t stream);
/* dbn == NULL in synthsr_bn_maxpool_bwd_ex: only the sums are produced.  synthsr_bn_pool_elu_bwd then does MaxPooling3D
 * backward + BatchNormalization backward + ELU backward of an encoder level in one pass (ext/neuron/models.py:316-356):
 * dz = (BN'(route(dpool)) + dy2) * ELU'(y) with y [shape][C] the conv + ELU output the BatchNorm read, dpool [shape/2][C] the
 * gradient w.r.t. the pooled tensor, dy2 (optional) the skip connection's gradient w.r.t. y, sums as above; dbias[C]
 * (optional) += sum over voxels of dz.  Bit-identical to bn_maxpool_bwd_ex + bn_elu_bwd, without writing and re-reading the
 * 7/8-zero routed gradient. */
int synthsr_bn_pool_elu_bwd(const float* dpool, const float* y, const float* dy2, float* dz, float* dbias, const int shape[3],
                            int C, const float* stats, const float* gamma, const float* beta, const float* sums, float eps,
                            synthsr_stream_t stream);
int synthsr_bn_pool_elu_bwd_bf16(const void* dpool, const void* y, const void* dy2, void* dz, float* dbias, const int shape[3],
                                 int C, const float* stats, const float* gamma, const float* beta, const float* sums, float eps,
                                 synthsr_stream_t stream);
/* BN backward, pass 1: sums[0..C) = sum dy, sums[C..2C) = sum dy*xhat (zeroed by caller) */
int synthsr_bn_bwd_reduce(const float* dy, const float* x, int64_t nvox, int C, const float* stats, float eps,
                          float* sums, synthsr_stream_t stream);
/* BN backward, pass 2: dx = gamma*invstd*(dy - sum_dy/n - xhat*sum_dyxhat/n); dgamma = sum_dyxhat; dbeta = sum_dy */
int synthsr_bn_bwd_apply(const float* dy, const float* x, float* dx, int64_t nvox, int C, const float* stats,
                         const float* gamma, float eps, const float* sums, synthsr_stream_t stream);

/* UpSampling3D(2) nearest of BN(lo) + concatenate([skip, up]) (models.py:426-434):
 * skip [d0,d1,d2,Cs], lo [d0/2,d1/2,d2/2,Cl] -> out [d0,d1,d2,Cs+Cl]; BN (stats,gamma,beta) applied to lo */
int synthsr_upsample_concat(const float* skip, const float* lo, float* out, const int shape[3], int Cs, int Cl,
                            const float* stats, const float* gamma, const float* beta, float eps,
                            synthsr_stream_t stream);
/* backward: dcat [d,Cs+Cl] -> dskip [d,Cs] (written), dlo_bn [d/2,Cl] = sum over the 2^3 children */
int synthsr_upsample_concat_bwd(const float* dcat, float* dskip, float* dlo_bn, const int shape[3], int Cs, int Cl,
                                synthsr_stream_t stream);

/* unet_likelihood: Conv3D(nb_labels=1, 1x1x1, linear) on BN(x) (models.py:480-481) fused with the
 * L1 loss of metrics_model (SynthSR/metrics_model.py:102-104):
 *   pred[v] = sum_c w[c]*bn(x[v][c]) + b (+ residual[v*res_stride+res_off] if residual != NULL)
 *   loss += sum_v |pred - target| / nvox        (loss: device float, zeroed by caller)
 *   dpred[v] = sign(pred-target)/nvox */
int synthsr_head_l1_fwd(const float* x, int64_t nvox, int C, const float* stats, const float* gamma,
                        const float* beta, float eps, const float* w, const float* b, const float* residual,
                        int res_stride, int res_off, const float* target, float* pred, float* dpred, float* loss,
                        synthsr_stream_t stream);
/* unet_likelihood with K output channels (w [C][K], b [K]) fused with the regression loss of metrics_model
 * (SynthSR/metrics_model.py:30-132), chosen by `kind` (training(regression_metric=...), SynthSR/training.py:85), for
 * n regression targets (training(output_channel=[...]); target [nvox][n]):
 *   0 'l1'  K=n  mean |pred - target|            1 'l2'  K=n  mean (pred - target)^2
 *   2 'laplace' K=2n (n intensity then n spread channels, training.py:325-326):
 *     mean( log(2b) + |pred_k - target_k| / b ),  b = 1e-5 + 0.02 exp(pred_{n+k})
 * (means over voxels and target channels; 1 <= K <= 4).  shape: the volume's 3 spatial sizes (host).  crop (host, NULL =
 * whole volume): {begin[3], size[3]} = the centred loss_cropping box (metrics_model.py:70-90): the mean runs over the
 * box, voxels outside get zero gradient.  residual (optional, [nvox][res_stride]): channel res_offs[k] (host, n entries)
 * is added to intensity channel k (work_with_residual_channel).  pred [nvox][K] (optional), dpred [nvox][K] (optional) =
 * dloss/dpred, loss: device float, zeroed by the caller. */
int synthsr_head_loss_fwd(const float* x, const int shape[3], int C, const float* stats, const float* gamma,
                          const float* beta, float eps, const float* w, const float* b, int K, const float* residual,
                          int res_stride, const int* res_offs, const float* target, float* pred, float* dpred,
                          float* loss, int kind, const int* crop, synthsr_stream_t stream);
/* synthsr_head_loss_fwd for ONE regression target with an l1 / l2 loss (kind 0 / 1), which also accumulates the two sums its
 * backward pass needs: ab[0 .. C) += sum_v g[v] xhat[v][c], ab[C] += sum_v g[v] (g = dpred; ab zeroed by the caller).  The
 * gradient w.r.t. the last BatchNorm's output is rank-1 (g[v] w[c]), so synthsr_head_bwd_from_sums(ab) yields dw (+=), db (+=)
 * and that BatchNorm's backward sums bn_sums [2C] (+=, may be NULL) without another pass over the feature map
 * (= synthsr_head_bwd_ex with dbn == NULL).  Only valid while dpred is not modified between the two calls. */
int synthsr_head_loss_fwd_ab(const float* x, const int shape[3], int C, const float* stats, const float* gamma, const float* beta,
                             float eps, const float* w, const float* b, const float* residual, int res_stride, int res_off,
                             const float* target, float* pred, float* dpred, float* loss, int kind, const int* crop, float* ab,
                             synthsr_stream_t stream);
int synthsr_head_loss_fwd_ab_bf16(const void* x, const int shape[3], int C, const float* stats, const float* gamma,
                                  const float* beta, float eps, const float* w, const float* b, const float* residual,
                                  int res_stride, int res_off, const float* target, float* pred, float* dpred, float* loss,
                                  int kind, const int* crop, float* ab, synthsr_stream_t stream);
int synthsr_head_bwd_from_sums(const float* ab, int C, const float* gamma, const float* beta, const float* w, float* dw,
                               float* db, float* bn_sums, synthsr_stream_t stream);
/* regression_metric='ssim' (SynthSR/metrics_model.py:105-125; tf.image.ssim(max_val=1): 11x11 Gaussian window sigma 1.5,
 * 'VALID', k1 .01, k2 .03).  Building blocks, orchestrated by synthsr_amd/ops.py:ssim_loss; all volumes planar float32.
 *   products: maps [4][box] = pred, target, pred*target, pred^2 + target^2 over the (loss_cropping) box `crop`
 *             ({begin[3], size[3]}, host; NULL = whole volume `shape`)
 *   filter:   11-tap correlation (host `taps`) along `axis` of `nmaps` volumes of `shape`; full = 0: 'VALID' (length - 10),
 *             full = 1: its transpose (length + 10, zero padded) used by the backward pass
 *   point:    filtered = [4][nq] (mu_x, mu_y, e_xy, e_2) -> *loss += scale * sum(luminance * cs); grads (optional) [3][nq]
 *             = scale * d(lum*cs)/d(mu_x, e_xy, e_2)
 *   combine:  dpred[box] += G0 + target*G1 + 2*pred*G2 with gback = [3][box] the gradient maps filtered back */
int synthsr_ssim_products(const float* pred, const float* target, const int shape[3], const int* crop, float* maps,
                          synthsr_stream_t stream);
int synthsr_ssim_filter(const float* in, float* out, const int shape[3], int axis, int full, int nmaps, const float* taps,
                        synthsr_stream_t stream);
int synthsr_ssim_point(const float* filtered, int64_t nq, float max_val, float scale, float* loss, float* grads,
                       synthsr_stream_t stream);
int synthsr_ssim_combine(const float* gback, const float* pred, const float* target, const int shape[3], const int* crop,
                         float* dpred, synthsr_stream_t stream);
/* backward of a K-channel head (2 <= K <= 4): dbn[v][c] = sum_k dpred[v][k]*w[c][k] (written), dw [C][K] +=, db [K] += */
int synthsr_head_bwd_multi(const float* dpred, const float* x, int64_t nvox, int C, int K, const float* stats,
                           const float* gamma, const float* beta, float eps, const float* w, float* dbn, float* dw,
                           float* db, synthsr_stream_t stream);
/* head backward: dbn[v][c] = dpred[v]*w[c]; dw[c] += sum_v dpred[v]*bn(x[v][c]); db += sum_v dpred[v] */
int synthsr_head_bwd(const float* dpred, const float* x, int64_t nvox, int C, const float* stats,
                     const float* gamma, const float* beta, float eps, const float* w, float* dbn, float* dw,
                     float* db, synthsr_stream_t stream);
/* as above with dbn optional (NULL: not written) and, when bn_sums != NULL, the BN-backward channel sums of that rank-1
 * gradient, bn_sums[c] += w[c] sum_v dpred,  bn_sums[C+c] += w[c] sum_v dpred*xhat[v][c]  (= synthsr_bn_bwd_reduce(dbn)) */
int synthsr_head_bwd_ex(const float* dpred, const float* x, int64_t nvox, int C, const float* stats,
                        const float* gamma, const float* beta, float eps, const float* w, float* dbn, float* dw,
                        float* db, float* bn_sums, synthsr_stream_t stream);

/* --- segmentation-regularised loss (SynthSR/metrics_model.py:136-215) -------------------------------------------------
 * head of the frozen segmentation U-Net: probs[v][n] = softmax_n(sum_c w[c][n]*bn(x[v][c]) + b[n]); C, N <= 64 */
int synthsr_seg_head_fwd(const float* x, int64_t nvox, int C, const float* stats, const float* gamma, const float* beta,
                         float eps, const float* w, const float* b, int N, float* probs, synthsr_stream_t stream);
/* soft-Dice sums (ext/lab2im/layers.py:1343-1362, enable_checks=False): class k merges the segmentation labels
 * cls_idx[k][0..2] (-1 = unused) and is compared with (seg == cls_gt[k]); sums[k] += 2 gt pred, sums[K+k] += gt^2 + pred^2
 * (sums zeroed by the caller); loss = mean_k(1 - (sums[k] + 1e-7)/(sums[K+k] + 1e-7)) */
int synthsr_seg_dice_sums(const float* probs, const int32_t* seg, int64_t nvox, int N, const int32_t* cls_idx,
                          const int32_t* cls_gt, int K, float* sums, synthsr_stream_t stream);
/* gradient of scale*loss w.r.t. the BatchNorm output in front of the head: dbn[v][c] (through label merging, softmax, 1x1x1) */
int synthsr_seg_dice_bwd(const float* probs, const int32_t* seg, int64_t nvox, int C, int N, const float* w,
                         const int32_t* cls_idx, const int32_t* cls_gt, int K, const float* sums, float scale, float* dbn,
                         synthsr_stream_t stream);

/* keras.optimizers.Adam (Keras 2.3.1; SynthSR/training.py:444): lr_t = lr*sqrt(1-b2^t)/(1-b1^t),
 * p -= lr_t*m/(sqrt(v)+eps).  lr already includes the 1/(1+decay*iter) factor. */
int synthsr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1,
                      float beta2, float eps, float grad_scale, synthsr_stream_t stream);

/* Stride-2 'same' Conv3D of an even-sized volume through the parity kernels of the folded decoder conv: pack the weights with
 * synthsr_conv3d_pack_ex(..., up = 2) on the LOW-RES (output) shape - mode 0 for the forward pass, evaluated by
 * synthsr_conv3d_up_dgrad(x_hi, packed, y_lo, lo_shape, Cl = Cout, Cout = Cin), mode 1 for the data gradient, evaluated by
 * synthsr_conv3d_up_fwd(dy_lo, packed, NULL, NULL, dx_hi, lo_shape, Cl = Cout, Cout = Cin, 0); weight gradient:
 * synthsr_conv3d_up_wgrad(dy_lo, x_hi, dwc, lo_shape, Cl = Cout, Cout = Cin) followed by this unpack
 * (dwc [8][27][Cout][Cin] zeroed by the caller, dw [27][Cin][Cout] +=). */
int synthsr_conv3d_stride_unpack(const float* dwc, float* dw, int Cin, int Cout, synthsr_stream_t stream);

/* ------------------------------------------------------------------ WGAN-GP critic pieces
 * (SynthSR/fine_tuning_with_adversary.py:482-508 `make_discriminator`, :579-595 `build_discriminator_loss`) */
/* LeakyReLU: dy == NULL: out = x > 0 ? x : alpha x (in place allowed); else out = dy * (x > 0 ? 1 : alpha), x = layer output */
int synthsr_leaky_relu(const float* x, const float* dy, float* out, int64_t n, float alpha, synthsr_stream_t stream);
/* out = LeakyReLU(x + bias[c]) for channels-last x (n values, C channels); out [C] += column sums of x [n][C] */
int synthsr_bias_leaky_relu(const float* x, const float* bias, float* out, int64_t n, int C, float alpha,
                            synthsr_stream_t stream);
int synthsr_colsum(const float* x, int64_t n, int C, float* out, synthsr_stream_t stream);
/* Dense: y [n_out] = b + x [n_in] . W [n_in][n_out] (Keras layout; n_out <= 1024; b optional) */
int synthsr_dense_fwd(const float* x, const float* W, const float* b, float* y, int64_t n_in, int n_out,
                      synthsr_stream_t stream);
/* dx [n_in] = W dy (optional, written), dW [n_in][n_out] += x (x) dy (optional) */
int synthsr_dense_bwd(const float* x, const float* W, const float* dy, float* dx, float* dW, int64_t n_in, int n_out,
                      synthsr_stream_t stream);
/* out = x * y ; out[i] = lut[labels[i]] (ConvertLabels, ext/lab2im/layers.py:1659-1689; 0 outside the table) */
int synthsr_mul(const float* x, const float* y, float* out, int64_t n, synthsr_stream_t stream);
int synthsr_lut_gather(const int* labels, const float* lut, int n_lut, float* out, int64_t n, synthsr_stream_t stream);
/* out = a x + b y (y optional) ; *out += sum x^2 */
int synthsr_axpby(const float* x, const float* y, float* out, int64_t n, float a, float b, synthsr_stream_t stream);
int synthsr_sumsq(const float* x, int64_t n, float* out, synthsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SYNTHSR_HIP_H */
