// Synthetic-brain generator kernels (gfx950).  HBM-bound gather / stencil / elementwise work:
// one thread per output voxel, coalesced along the fastest spatial axis, no MFMA.
//
// THIS FILE MUST BE COMPILED WITH -ffp-contract=off: label indexing has to be bit-exact with the
// oracle, whose float32 coordinate pipeline rounds after every multiply and every add
// (reference op order: ext/neuron/utils.py:68-110, 150, 271-286, 316-317).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------
// edge-clamped trilinear setup for one axis — ext/neuron/utils.py:68-84
struct Axis {
  int i0, i1;
  float w0, w1;  // weight of the floor corner (diff_loc1) and of the ceil corner (1 - diff_loc1)
};

__device__ __forceinline__ Axis axis_setup(float loc, int n) {
  Axis a;
  const float mx = (float)(n - 1);
  const float fl = floorf(loc);
  const float cl = fminf(fmaxf(loc, 0.f), mx);
  const float f0 = fminf(fmaxf(fl, 0.f), mx);
  const float f1 = fminf(fmaxf(f0 + 1.f, 0.f), mx);
  a.w0 = f1 - cl;
  a.w1 = 1.f - a.w0;
  a.i0 = (int)f0;
  a.i1 = (int)f1;
  return a;
}

// value of channel c at the 8 corners, accumulated in the reference's corner order
// itertools.product([0,1], repeat=3) with weight (w[c0][0]*w[c1][1])*w[c2][2]  (utils.py:88-110, 530-534)
template <typename F>
__device__ __forceinline__ float tri_accum(const Axis& a0, const Axis& a1, const Axis& a2, F fetch) {
  float acc;
  {
    const float w = (a0.w0 * a1.w0) * a2.w0;
    acc = w * fetch(a0.i0, a1.i0, a2.i0);
  }
  acc = acc + ((a0.w0 * a1.w0) * a2.w1) * fetch(a0.i0, a1.i0, a2.i1);
  acc = acc + ((a0.w0 * a1.w1) * a2.w0) * fetch(a0.i0, a1.i1, a2.i0);
  acc = acc + ((a0.w0 * a1.w1) * a2.w1) * fetch(a0.i0, a1.i1, a2.i1);
  acc = acc + ((a0.w1 * a1.w0) * a2.w0) * fetch(a0.i1, a1.i0, a2.i0);
  acc = acc + ((a0.w1 * a1.w0) * a2.w1) * fetch(a0.i1, a1.i0, a2.i1);
  acc = acc + ((a0.w1 * a1.w1) * a2.w0) * fetch(a0.i1, a1.i1, a2.i0);
  acc = acc + ((a0.w1 * a1.w1) * a2.w1) * fetch(a0.i1, a1.i1, a2.i1);
  return acc;
}

// the same for a 3-component field stored interleaved ([...][3]): ONE 12-byte load per corner instead of three 4-byte
// gathers (the address unit handles a gather lane by lane: 8 instead of 24 gather instructions per voxel); per component
// the arithmetic is tri_accum's, term by term
struct F3 {
  float x, y, z;
};
template <typename F>
__device__ __forceinline__ void tri_accum3(const Axis& a0, const Axis& a1, const Axis& a2, F fetch, float& r0, float& r1,
                                           float& r2) {
  float c0, c1, c2;
  {
    const float w = (a0.w0 * a1.w0) * a2.w0;
    const F3 f = fetch(a0.i0, a1.i0, a2.i0);
    c0 = w * f.x;
    c1 = w * f.y;
    c2 = w * f.z;
  }
#define SYN_TRI3(W, I, J, K)          \
  {                                   \
    const float w = (W);              \
    const F3 f = fetch((I), (J), (K)); \
    c0 = c0 + w * f.x;                \
    c1 = c1 + w * f.y;                \
    c2 = c2 + w * f.z;                \
  }
  SYN_TRI3((a0.w0 * a1.w0) * a2.w1, a0.i0, a1.i0, a2.i1)
  SYN_TRI3((a0.w0 * a1.w1) * a2.w0, a0.i0, a1.i1, a2.i0)
  SYN_TRI3((a0.w0 * a1.w1) * a2.w1, a0.i0, a1.i1, a2.i1)
  SYN_TRI3((a0.w1 * a1.w0) * a2.w0, a0.i1, a1.i0, a2.i0)
  SYN_TRI3((a0.w1 * a1.w0) * a2.w1, a0.i1, a1.i0, a2.i1)
  SYN_TRI3((a0.w1 * a1.w1) * a2.w0, a0.i1, a1.i1, a2.i0)
  SYN_TRI3((a0.w1 * a1.w1) * a2.w1, a0.i1, a1.i1, a2.i1)
#undef SYN_TRI3
  r0 = c0;
  r1 = c1;
  r2 = c2;
}

// resize sample position: i + (i/zoom - i)  (utils.py:150 then :317)
__device__ __forceinline__ float resize_pos(int i, float zoom) {
  const float g = (float)i;
  return g + (g / zoom - g);
}

struct Shape3 {
  int d[3];
};

// flat voxel index -> (i0, i1, i2).  Two 32-bit divisions (a 64-bit division by a run-time divisor is a ~100-instruction
// sequence on gfx950 and every generator kernel did three per voxel); volumes of 2^31 voxels and more take the 64-bit path
__device__ __forceinline__ void vox3(int64_t v, int d1, int d2, int& i0, int& i1, int& i2) {
  if (v >> 31) {
    i2 = (int)(v % d2);
    i1 = (int)((v / d2) % d1);
    i0 = (int)(v / ((int64_t)d2 * d1));
    return;
  }
  const uint32_t u = (uint32_t)v, q = u / (uint32_t)d2, q1 = q / (uint32_t)d1;
  i2 = (int)(u - q * (uint32_t)d2);
  i1 = (int)(q - q1 * (uint32_t)d1);
  i0 = (int)q1;
}

// ------------------------------------------------------------------------------------------
__global__ void resize_kernel(const float* __restrict__ in, float* __restrict__ out, int C, Shape3 is, Shape3 os,
                              float z0, float z1, float z2, int method) {
  const int64_t n = (int64_t)os.d[0] * os.d[1] * os.d[2];
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    int o0, o1, o2;
    vox3(v, os.d[1], os.d[2], o0, o1, o2);
    const float l0 = resize_pos(o0, z0), l1 = resize_pos(o1, z1), l2 = resize_pos(o2, z2);
    if (method == 1) {
      int r0 = (int)rintf(l0), r1 = (int)rintf(l1), r2 = (int)rintf(l2);
      r0 = min(max(r0, 0), is.d[0] - 1);
      r1 = min(max(r1, 0), is.d[1] - 1);
      r2 = min(max(r2, 0), is.d[2] - 1);
      const float* src = in + (((int64_t)r0 * is.d[1] + r1) * is.d[2] + r2) * C;
      for (int c = 0; c < C; ++c) out[v * C + c] = src[c];
    } else {
      const Axis a0 = axis_setup(l0, is.d[0]), a1 = axis_setup(l1, is.d[1]), a2 = axis_setup(l2, is.d[2]);
      for (int c = 0; c < C; ++c) {
        out[v * C + c] = tri_accum(a0, a1, a2, [&](int i, int j, int k) {
          return in[(((int64_t)i * is.d[1] + j) * is.d[2] + k) * C + c];
        });
      }
    }
  }
}

// one scaling-and-squaring step: out = s*v + interp(s*v, x + s*v)   (utils.py:366-369; s = 2^-n only on step 0)
__global__ void svf_step_kernel(const float* __restrict__ in, float* __restrict__ out, Shape3 s, float scale) {
  const int64_t n = (int64_t)s.d[0] * s.d[1] * s.d[2];
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    int i0, i1, i2;
    vox3(v, s.d[1], s.d[2], i0, i1, i2);
    const float v0 = in[v * 3 + 0] * scale, v1 = in[v * 3 + 1] * scale, v2 = in[v * 3 + 2] * scale;
    const Axis a0 = axis_setup((float)i0 + v0, s.d[0]);
    const Axis a1 = axis_setup((float)i1 + v1, s.d[1]);
    const Axis a2 = axis_setup((float)i2 + v2, s.d[2]);
    float r[3];
    tri_accum3(a0, a1, a2, [&](int i, int j, int k) {
      const F3 f = *reinterpret_cast<const F3*>(in + (((int64_t)i * s.d[1] + j) * s.d[2] + k) * 3);
      return F3{f.x * scale, f.y * scale, f.z * scale};
    }, r[0], r[1], r[2]);
    out[v * 3 + 0] = v0 + r[0];
    out[v * 3 + 1] = v1 + r[1];
    out[v * 3 + 2] = v2 + r[2];
  }
}

struct Aff {
  float a[12];
};

// sampling position of voxel (i0,i1,i2): x + (A.[x_c + u; 1] - x_c)   (utils.py:271-286, 316-317)
__device__ __forceinline__ void affine_pos(const Aff& A, const int S[3], int i0, int i1, int i2, float u0, float u1,
                                           float u2, bool has_u, bool has_aff, float pos[3]) {
  if (!has_aff) {  // single dense transform: shift = u
    pos[0] = (float)i0 + u0;
    pos[1] = (float)i1 + u1;
    pos[2] = (float)i2 + u2;
    return;
  }
  const float c0 = (float)i0 - (float)((S[0] - 1) / 2.0);
  const float c1 = (float)i1 - (float)((S[1] - 1) / 2.0);
  const float c2 = (float)i2 - (float)((S[2] - 1) / 2.0);
  const float m0 = has_u ? c0 + u0 : c0;
  const float m1 = has_u ? c1 + u1 : c1;
  const float m2 = has_u ? c2 + u2 : c2;
  const float l0 = ((A.a[0] * m0 + A.a[1] * m1) + A.a[2] * m2) + A.a[3];
  const float l1 = ((A.a[4] * m0 + A.a[5] * m1) + A.a[6] * m2) + A.a[7];
  const float l2 = ((A.a[8] * m0 + A.a[9] * m1) + A.a[10] * m2) + A.a[11];
  pos[0] = (float)i0 + (l0 - c0);
  pos[1] = (float)i1 + (l1 - c1);
  pos[2] = (float)i2 + (l2 - c2);
}

__global__ void affine_resample_kernel(const float* __restrict__ in, float* __restrict__ out, int C, Shape3 s,
                                       Aff A) {
  const int64_t n = (int64_t)s.d[0] * s.d[1] * s.d[2];
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    int i0, i1, i2;
    vox3(v, s.d[1], s.d[2], i0, i1, i2);
    float pos[3];
    affine_pos(A, s.d, i0, i1, i2, 0.f, 0.f, 0.f, false, true, pos);
    const Axis a0 = axis_setup(pos[0], s.d[0]), a1 = axis_setup(pos[1], s.d[1]), a2 = axis_setup(pos[2], s.d[2]);
    for (int c = 0; c < C; ++c) {
      out[v * C + c] = tri_accum(a0, a1, a2, [&](int i, int j, int k) {
        return in[(((int64_t)i * s.d[1] + j) * s.d[2] + k) * C + c];
      });
    }
  }
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), counter = (voxel_lo, voxel_hi, offset_lo, offset_hi)
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0;
    c[1] = lo1;
    c[2] = n2;
    c[3] = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

__device__ __forceinline__ void box_muller(uint32_t r0, uint32_t r1, float& n0, float& n1) {
  const float u1 = (float)((r0 >> 8) + 1u) * 5.9604644775390625e-08f;  // (0, 1]
  const float u2 = (float)(r1 >> 8) * 5.9604644775390625e-08f;         // [0, 1)
  // hardware transcendentals: v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32 take their argument in REVOLUTIONS, i.e.
  // u2 itself -- no 2 pi multiply, no range reduction (the precise libm sinf / cosf / logf were ~150 instructions per
  // voxel).  This is the in-kernel sampling path (use_philox); parity runs feed the noise from a tape.
  const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // -2 ln(u1) = -2 ln2 log2(u1)
  n0 = rad * __builtin_amdgcn_cosf(u2);
  n1 = rad * __builtin_amdgcn_sinf(u2);
}

// ------------------------------------------------------------------------------------------
// fused: full-res field (trilinear from the half-res SVF) -> affine∘elastic position -> nearest label
// -> [crop, flip, L/R swap] -> GMM -> [bias] -> [clip] -> planar channel + block min/max
__global__ __launch_bounds__(256) void deform_gmm_kernel(const int32_t* __restrict__ labels,
                                                         const float* __restrict__ field,
                                                         const float* __restrict__ gmm_lut,
                                                         const int32_t* __restrict__ swap_lut,
                                                         const float* __restrict__ noise,
                                                         const float* __restrict__ bias_small,
                                                         int32_t* __restrict__ seg_out, float* __restrict__ chan_out,
                                                         uint32_t* __restrict__ minmax, const float* __restrict__ real_in,
                                                         float* __restrict__ real_out, uint32_t* __restrict__ real_minmax,
                                                         synthsr_deform_params p) {
  const int64_t n = (int64_t)p.out_shape[0] * p.out_shape[1] * p.out_shape[2];
  const int C = p.n_channels;
  // more than four synthetic channels: one launch per group of four (chan_first = 4 k); Ct = channels of the whole model (the
  // rows of the GMM LUT and of a noise tape), 0 = this launch has them all
  const int Ct = p.n_channels_total > 0 ? p.n_channels_total : C, c00 = p.n_channels_total > 0 ? p.chan_first : 0;
  float lmin[4], lmax[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    lmin[c] = INFINITY;
    lmax[c] = -INFINITY;
  }
  float rmin = INFINITY, rmax = -INFINITY;  // real-image regression target (labels_to_image_model.py:126-134)
  Aff A;
#pragma unroll
  for (int i = 0; i < 12; ++i) A.a[i] = p.aff[i];
  const float zf0 = p.has_field ? (float)((double)p.in_shape[0] / (double)p.half_shape[0]) : 1.f;
  const float zf1 = p.has_field ? (float)((double)p.in_shape[1] / (double)p.half_shape[1]) : 1.f;
  const float zf2 = p.has_field ? (float)((double)p.in_shape[2] / (double)p.half_shape[2]) : 1.f;

  int64_t o_lo, o_hi;
  syn_block_range(n, o_lo, o_hi);  // XCD-contiguous slabs: the label gather stays in one L2
  for (int64_t o = o_lo + threadIdx.x; o < o_hi; o += blockDim.x) {
    int o0, o1, o2;
    vox3(o, p.out_shape[1], p.out_shape[2], o0, o1, o2);
    // undo flip (tf.reverse along axis 0), then crop offset -> voxel of the deformed full grid
    const int i0 = (p.flip ? (p.out_shape[0] - 1 - o0) : o0) + p.crop[0];
    const int i1 = o1 + p.crop[1];
    const int i2 = o2 + p.crop[2];
    float u0 = 0.f, u1 = 0.f, u2 = 0.f;
    if (p.has_field) {  // Resize(linear) of the integrated SVF to full size (layers.py:196), evaluated on the fly
      const Axis a0 = axis_setup(resize_pos(i0, zf0), p.half_shape[0]);
      const Axis a1 = axis_setup(resize_pos(i1, zf1), p.half_shape[1]);
      const Axis a2 = axis_setup(resize_pos(i2, zf2), p.half_shape[2]);
      const int h1 = p.half_shape[1], h2 = p.half_shape[2];
      tri_accum3(a0, a1, a2, [&](int i, int j, int k) {
        return *reinterpret_cast<const F3*>(field + ((i * h1 + j) * h2 + k) * 3);  // half-res field: < 2^31 elements
      }, u0, u1, u2);
    }
    float pos[3];
    affine_pos(A, p.in_shape, i0, i1, i2, u0, u1, u2, p.has_field != 0, p.has_affine != 0, pos);
    int r0 = (int)rintf(pos[0]), r1 = (int)rintf(pos[1]), r2 = (int)rintf(pos[2]);  // tf.round: half to even
    r0 = min(max(r0, 0), p.in_shape[0] - 1);
    r1 = min(max(r1, 0), p.in_shape[1] - 1);
    r2 = min(max(r2, 0), p.in_shape[2] - 1);
    const int64_t li = ((int64_t)r0 * p.in_shape[1] + r1) * p.in_shape[2] + r2;
    // the label volume as uint8 / int16 when its values fit (the resident training pool: 1-2 instead of 4 bytes per gathered voxel)
    int lab = p.label_bytes == 1 ? (int)reinterpret_cast<const uint8_t*>(labels)[li]
                                 : (p.label_bytes == 2 ? (int)reinterpret_cast<const int16_t*>(labels)[li] : labels[li]);
    if (p.flip && swap_lut != nullptr && lab >= 0 && lab < p.swap_lut_size) lab = swap_lut[lab];
    if (seg_out) seg_out[o] = lab;
    if (real_in) {  // the same transform with inter_method 'linear' (edge-clamped trilinear, ext/neuron/utils.py:67-110)
      const Axis r0a = axis_setup(pos[0], p.in_shape[0]);
      const Axis r1a = axis_setup(pos[1], p.in_shape[1]);
      const Axis r2a = axis_setup(pos[2], p.in_shape[2]);
      const int s1 = p.in_shape[1], s2 = p.in_shape[2];
      const float rv = tri_accum(r0a, r1a, r2a, [&](int i, int j, int k) { return real_in[((int64_t)i * s1 + j) * s2 + k]; });
      real_out[o] = rv;
      rmin = fminf(rmin, rv);
      rmax = fmaxf(rmax, rv);
    }

    float nz[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.use_philox) {
      uint32_t c[4] = {(uint32_t)o, (uint32_t)((uint64_t)o >> 32), (uint32_t)p.philox_offset,
                       (uint32_t)(p.philox_offset >> 32) + ((uint32_t)(c00 >> 2) << 16)};  // channel group in the counter
      philox4x32_10(c, p.philox_key[0], p.philox_key[1]);
      box_muller(c[0], c[1], nz[0], nz[1]);
      if (C > 2) box_muller(c[2], c[3], nz[2], nz[3]);
    } else {
      for (int c = 0; c < C; ++c) nz[c] = noise[o * Ct + c00 + c];
    }
    const bool known = (lab >= 0 && lab < p.lut_size);
    int boff = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < C) {
        const float mu = known ? gmm_lut[(c00 + c) * p.lut_size + lab] : 0.f;
        const float sd = known ? gmm_lut[(Ct + c00 + c) * p.lut_size + lab] : 0.f;
        float x = sd * nz[c] + mu;  // layers.py:498
        const int b0 = p.bias_shape[c][0], b1 = p.bias_shape[c][1], b2 = p.bias_shape[c][2];
        if (b0 > 0) {
          if (p.bias_on[c]) {  // layers.py:1083-1088 (field lives on the OUTPUT grid)
            const float z0 = (float)((double)p.out_shape[0] / (double)b0);
            const float z1 = (float)((double)p.out_shape[1] / (double)b1);
            const float z2 = (float)((double)p.out_shape[2] / (double)b2);
            const Axis a0 = axis_setup(resize_pos(o0, z0), b0);
            const Axis a1 = axis_setup(resize_pos(o1, z1), b1);
            const Axis a2 = axis_setup(resize_pos(o2, z2), b2);
            const float* bs = bias_small + boff;
            const float b = tri_accum(a0, a1, a2, [&](int i, int j, int k) { return bs[(i * b1 + j) * b2 + k]; });
            x = expf(b) * x;
          }
          boff += b0 * b1 * b2;
        }
        if (p.clip_hi > 0.f) {
          x = fminf(fmaxf(x, 0.f), p.clip_hi);
          if (x == 0.f) x = 0.f;  // canonical +0
        }
        chan_out[(int64_t)c * n + o] = x;
        lmin[c] = fminf(lmin[c], x);
        lmax[c] = fmaxf(lmax[c], x);
      }
    }
  }
  // block min/max -> one atomic pair per wave per channel
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (c < C) {
      const float mn = syn_wave_min(lmin[c]), mx = syn_wave_max(lmax[c]);
      if ((threadIdx.x & 63) == 0) syn_minmax_update(&minmax[2 * c], mn, mx);
    }
  }
  if (real_in && real_minmax) {
    const float mn = syn_wave_min(rmin), mx = syn_wave_max(rmax);
    if ((threadIdx.x & 63) == 0) syn_minmax_update(real_minmax, mn, mx);
  }
}

// MimicAcquisition (ext/lab2im/layers.py:927-990) with min_subsample_res = volume_res and a distance map: nearest-neighbour
// down-sampling and linear up-sampling are composed per output voxel (the intermediate low-resolution tensor is never
// written): corner j of the trilinear stencil on the (full-size, edge-replicated) down-sampled grid reads the source
// voxel clip(rint(clip(j / down_zoom, 0, S)), 0, S-1).  dist = | min(frac, 1-frac) * subsample_res |_2.
struct MimicP {
  int S[3], R[3];
  float dz[3], uz[3], sub[3];
};
__global__ __launch_bounds__(256) void mimic_acquisition_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                MimicP p, int ostride, int ooff, int doff) {
  const int64_t n = (int64_t)p.R[0] * p.R[1] * p.R[2];
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < n; o += (int64_t)gridDim.x * blockDim.x) {
    int o0, o1, o2;
    vox3(o, p.R[1], p.R[2], o0, o1, o2);
    const float u0 = (float)o0 / p.uz[0], u1 = (float)o1 / p.uz[1], u2 = (float)o2 / p.uz[2];
    const Axis a0 = axis_setup(u0, p.S[0]);
    const Axis a1 = axis_setup(u1, p.S[1]);
    const Axis a2 = axis_setup(u2, p.S[2]);
    auto src = [&](int j, int d) {  // down-sampled index -> source index (nearest, both clamps of the reference)
      const float loc = fminf(fmaxf((float)j / p.dz[d], 0.f), (float)p.S[d]);
      return min(max((int)rintf(loc), 0), p.S[d] - 1);
    };
    const int s1 = p.S[1], s2 = p.S[2];
    const float v = tri_accum(a0, a1, a2, [&](int i, int j, int k) {
      return in[((int64_t)src(i, 0) * s1 + src(j, 1)) * s2 + src(k, 2)];
    });
    if (ooff >= 0) out[o * ostride + ooff] = v;
    if (doff >= 0) {
      const float d0 = fminf(u0 - floorf(u0), ceilf(u0) - u0) * p.sub[0];
      const float d1 = fminf(u1 - floorf(u1), ceilf(u1) - u1) * p.sub[1];
      const float d2 = fminf(u2 - floorf(u2), ceilf(u2) - u2) * p.sub[2];
      out[o * ostride + doff] = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
    }
  }
}

__global__ void minmax_init_kernel(uint32_t* mm, int n_pairs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pairs) {
    mm[2 * i + 0] = 0xFFFFFFFFu;
    mm[2 * i + 1] = 0u;
  }
}

__global__ void minmax_reduce_kernel(const float* __restrict__ x, int64_t n, uint32_t* __restrict__ mm) {
  float mn = INFINITY, mx = -INFINITY;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
  mn = syn_wave_min(mn);
  mx = syn_wave_max(mx);
  if ((threadIdx.x & 63) == 0) syn_minmax_update(mm, mn, mx);
}

__global__ void normalise_gamma_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n,
                                       const uint32_t* __restrict__ mm, float gexp) {
  const float m = syn_ord2f(mm[0]), M = syn_ord2f(mm[1]);
  const float den = (M - m) + 1e-7f;  // K.epsilon(), layers.py:1236
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = fminf(fmaxf(x[i], m), M);
    v = (v - m) / den;
    if (gexp > 0.f) v = powf(v, gexp);
    out[i] = v;
  }
}

// zero-padded cross-correlation, float32 accumulation in (z,y,x) raster order of the taps
__global__ __launch_bounds__(256) void blur3d_kernel(const float* __restrict__ in, float* __restrict__ out, Shape3 s,
                                                     const float* __restrict__ kern, Shape3 ks, int ostride,
                                                     int ooff, int foff, float fval) {
  __shared__ float kw[512];
  const int nk = ks.d[0] * ks.d[1] * ks.d[2];
  for (int i = threadIdx.x; i < nk; i += blockDim.x) kw[i] = kern[i];
  __syncthreads();
  const int64_t n = (int64_t)s.d[0] * s.d[1] * s.d[2];
  const int p0 = ks.d[0] / 2, p1 = ks.d[1] / 2, p2 = ks.d[2] / 2;
  int64_t v_lo, v_hi;
  syn_block_range(n, v_lo, v_hi);  // XCD-contiguous slabs: the 27-tap stencil re-reads hit this XCD's L2
  for (int64_t v = v_lo + threadIdx.x; v < v_hi; v += blockDim.x) {
    int i0, i1, i2;
    vox3(v, s.d[1], s.d[2], i0, i1, i2);
    float acc = 0.f;
    for (int a = 0; a < ks.d[0]; ++a) {
      const int z = i0 + a - p0;
      for (int b = 0; b < ks.d[1]; ++b) {
        const int y = i1 + b - p1;
        for (int c = 0; c < ks.d[2]; ++c) {
          const int x = i2 + c - p2;
          const bool ok = (z >= 0) & (z < s.d[0]) & (y >= 0) & (y < s.d[1]) & (x >= 0) & (x < s.d[2]);
          const float val = ok ? in[((int64_t)z * s.d[1] + y) * s.d[2] + x] : 0.f;
          acc = acc + val * kw[(a * ks.d[1] + b) * ks.d[2] + c];
        }
      }
    }
    out[v * ostride + ooff] = acc;
    if (foff >= 0) out[v * ostride + foff] = fval;
  }
}

// IntensityAugmentation's normalise + gamma (layers.py:1227-1242), GaussianBlur(sigma = .5) (labels_to_image_model.py:186:
// the regression-target tap) and the acquisition blur (:223) of ONE channel in a single pass: the clipped channel is read once
// with a 2-voxel halo into LDS (normalised on the way in), the sigma = .5 blur is evaluated on the tile + 1-voxel halo (its
// interior goes to the target tensor), the second blur on the tile itself (interleaved image + optional all-ones reliability
// map).  The three separate kernels moved 28 B / voxel through HBM (write + re-read of two intermediates), this one 16.
// Arithmetic is exactly that of normalise_gamma_kernel and blur3d_kernel: zero padding outside the VOLUME for both blurs (the
// sigma = .5 result outside the volume counts as 0 for the second blur, it is not evaluated there), taps accumulated in
// (z, y, x) raster order, float32, no contraction (this file is built with -ffp-contract=off).
constexpr int NB_TZ = 8, NB_TY = 16, NB_TX = 32;
__global__ __launch_bounds__(256) void normalise_blur2_kernel(const float* __restrict__ x, Shape3 s,
                                                              const uint32_t* __restrict__ mm, float gexp,
                                                              const float* __restrict__ k1, const float* __restrict__ k2,
                                                              float* __restrict__ target, float* __restrict__ image,
                                                              int istride, int ioff, int foff, float fval, int t1, int t2) {
  constexpr int AZ = NB_TZ + 4, AY = NB_TY + 4, AX = NB_TX + 4, BZ = NB_TZ + 2, BY = NB_TY + 2, BX = NB_TX + 2;
  __shared__ float A[AZ * AY * AX];
  __shared__ float B[BZ * BY * BX];
  __shared__ float kw[54];
  const int tid = threadIdx.x;
  if (tid < 54) kw[tid] = tid < 27 ? k1[tid] : k2[tid - 27];
  const float m = syn_ord2f(mm[0]), M = syn_ord2f(mm[1]);
  const float den = (M - m) + 1e-7f;  // K.epsilon(), layers.py:1236
  // XCD-contiguous tile order (consecutive workgroup ids are dealt to the 8 XCDs): each XCD walks a compact slab
  const int G = (int)gridDim.x, b = (int)blockIdx.x;
  const int tile = (G % 8 == 0) ? (b & 7) * (G >> 3) + (b >> 3) : b;
  const int tz = tile / (t1 * t2), r = tile - tz * (t1 * t2), ty = r / t2, tx = r - ty * t2;
  const int z0 = tz * NB_TZ, y0 = ty * NB_TY, x0 = tx * NB_TX;
  const int D0 = s.d[0], D1 = s.d[1], D2 = s.d[2];
  for (int i = tid; i < AZ * AY * AX; i += 256) {
    const int az = i / (AY * AX), q = i - az * (AY * AX), ay = q / AX, ax = q - ay * AX;
    const int z = z0 - 2 + az, y = y0 - 2 + ay, xx = x0 - 2 + ax;
    float v = 0.f;
    if ((unsigned)z < (unsigned)D0 && (unsigned)y < (unsigned)D1 && (unsigned)xx < (unsigned)D2) {
      v = fminf(fmaxf(x[((int64_t)z * D1 + y) * D2 + xx], m), M);
      v = (v - m) / den;
      if (gexp > 0.f) v = powf(v, gexp);
    }
    A[i] = v;
  }
  __syncthreads();
  for (int i = tid; i < BZ * BY * BX; i += 256) {
    const int bz = i / (BY * BX), q = i - bz * (BY * BX), by = q / BX, bx = q - by * BX;
    const int z = z0 - 1 + bz, y = y0 - 1 + by, xx = x0 - 1 + bx;
    float acc = 0.f;
    const bool inside = (unsigned)z < (unsigned)D0 && (unsigned)y < (unsigned)D1 && (unsigned)xx < (unsigned)D2;
    if (inside) {
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int d = 0; d < 3; ++d) acc = acc + A[((bz + a) * AY + (by + c)) * AX + (bx + d)] * kw[(a * 3 + c) * 3 + d];
      if (bz >= 1 && bz <= NB_TZ && by >= 1 && by <= NB_TY && bx >= 1 && bx <= NB_TX)
        target[((int64_t)z * D1 + y) * D2 + xx] = acc;
    }
    B[i] = acc;
  }
  __syncthreads();
  for (int i = tid; i < NB_TZ * NB_TY * NB_TX; i += 256) {
    const int cz = i / (NB_TY * NB_TX), q = i - cz * (NB_TY * NB_TX), cy = q / NB_TX, cx = q - cy * NB_TX;
    const int z = z0 + cz, y = y0 + cy, xx = x0 + cx;
    if (z < D0 && y < D1 && xx < D2) {
      float acc = 0.f;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int d = 0; d < 3; ++d) acc = acc + B[((cz + a) * BY + (cy + c)) * BX + (cx + d)] * kw[27 + (a * 3 + c) * 3 + d];
      const int64_t v = ((int64_t)z * D1 + y) * D2 + xx;
      image[v * istride + ioff] = acc;
      if (foff >= 0) image[v * istride + foff] = fval;
    }
  }
}

__global__ void outer3_kernel(const float* __restrict__ w, float* __restrict__ out, Shape3 s, int ostride, int ooff) {
  const int64_t n = (int64_t)s.d[0] * s.d[1] * s.d[2];
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    int i0, i1, i2;
    vox3(v, s.d[1], s.d[2], i0, i1, i2);
    // the reference multiplies the three 1-D maps in float64 and casts once (edit_tensors.py:326-328)
    const double r = ((double)w[i0] * (double)w[s.d[0] + i1]) * (double)w[s.d[0] + s.d[1] + i2];
    out[v * ostride + ooff] = (float)r;
  }
}

__global__ void copy_strided_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n, int is, int io,
                                    int os, int oo) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i * os + oo] = in[i * is + io];
}

inline bool bad_shape(const int s[3]) { return s[0] <= 0 || s[1] <= 0 || s[2] <= 0; }

}  // namespace

extern "C" {

int synthsr_resize_f32(const float* in, float* out, int C, const int in_shape[3], const int out_shape[3], int method,
                       synthsr_stream_t stream) {
  if (!in || !out || C <= 0 || bad_shape(in_shape) || bad_shape(out_shape) || (method != 0 && method != 1))
    return SYNTHSR_EINVAL;
  Shape3 is{{in_shape[0], in_shape[1], in_shape[2]}}, os{{out_shape[0], out_shape[1], out_shape[2]}};
  const float z0 = (float)((double)out_shape[0] / (double)in_shape[0]);
  const float z1 = (float)((double)out_shape[1] / (double)in_shape[1]);
  const float z2 = (float)((double)out_shape[2] / (double)in_shape[2]);
  const int64_t n = (int64_t)os.d[0] * os.d[1] * os.d[2];
  hipLaunchKernelGGL(resize_kernel, dim3(syn_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, C, is, os, z0,
                     z1, z2, method);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_svf_integrate(float* v, float* scratch, const int shape[3], int nb_steps, synthsr_stream_t stream) {
  if (!v || !scratch || bad_shape(shape) || nb_steps < 0 || nb_steps > 30) return SYNTHSR_EINVAL;
  Shape3 s{{shape[0], shape[1], shape[2]}};
  const int64_t n = (int64_t)s.d[0] * s.d[1] * s.d[2];
  const float scale = 1.0f / (float)(1u << nb_steps);
  float* a = v;
  float* b = scratch;
  if (nb_steps == 0) return SYNTHSR_OK;
  for (int i = 0; i < nb_steps; ++i) {
    hipLaunchKernelGGL(svf_step_kernel, dim3(syn_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, s,
                       i == 0 ? scale : 1.0f);
    SYN_CHECK_LAUNCH();
    float* t = a;
    a = b;
    b = t;
  }
  if (a != v) {
    if (hipMemcpyAsync(v, a, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) !=
        hipSuccess)
      return SYNTHSR_ELAUNCH;
  }
  return SYNTHSR_OK;
}

int synthsr_affine_resample_linear(const float* in, float* out, int C, const int shape[3], const float aff[12],
                                   synthsr_stream_t stream) {
  if (!in || !out || !aff || C <= 0 || bad_shape(shape)) return SYNTHSR_EINVAL;
  Shape3 s{{shape[0], shape[1], shape[2]}};
  Aff A;
  for (int i = 0; i < 12; ++i) A.a[i] = aff[i];
  const int64_t n = (int64_t)s.d[0] * s.d[1] * s.d[2];
  hipLaunchKernelGGL(affine_resample_kernel, dim3(syn_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, C, s,
                     A);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_deform_gmm(const int32_t* labels, const float* field_half, const float* gmm_lut, const int32_t* swap_lut,
                       const float* noise, const float* bias_small, int32_t* seg_out, float* chan_out,
                       uint32_t* minmax, const synthsr_deform_params* p, synthsr_stream_t stream) {
  return synthsr_deform_gmm_real(labels, field_half, gmm_lut, swap_lut, noise, bias_small, seg_out, chan_out, minmax,
                                 nullptr, nullptr, nullptr, p, stream);
}

int synthsr_deform_gmm_real(const int32_t* labels, const float* field_half, const float* gmm_lut,
                            const int32_t* swap_lut, const float* noise, const float* bias_small, int32_t* seg_out,
                            float* chan_out, uint32_t* minmax, const float* real_in, float* real_out,
                            uint32_t* real_minmax, const synthsr_deform_params* p, synthsr_stream_t stream) {
  if (!labels || !gmm_lut || !chan_out || !minmax || !p) return SYNTHSR_EINVAL;
  if (real_in && !real_out) return SYNTHSR_EINVAL;
  if (bad_shape(p->in_shape) || bad_shape(p->out_shape)) return SYNTHSR_EINVAL;
  if (p->n_channels < 1 || p->n_channels > 4 || p->lut_size < 1) return SYNTHSR_EINVAL;
  if (p->label_bytes != 0 && p->label_bytes != 1 && p->label_bytes != 2 && p->label_bytes != 4) return SYNTHSR_EINVAL;
  if (p->n_channels_total != 0 && (p->chan_first < 0 || (p->chan_first & 3) || p->chan_first + p->n_channels > p->n_channels_total))
    return SYNTHSR_EINVAL;
  if (p->has_field && (!field_half || bad_shape(p->half_shape))) return SYNTHSR_EINVAL;
  if (!p->use_philox && !noise) return SYNTHSR_EINVAL;
  if (p->swap_lut_size > 0 && !swap_lut) return SYNTHSR_EINVAL;
  for (int d = 0; d < 3; ++d)
    if (p->crop[d] < 0 || p->crop[d] + p->out_shape[d] > p->in_shape[d]) return SYNTHSR_EINVAL;
  for (int c = 0; c < p->n_channels; ++c) {
    if (p->bias_shape[c][0] > 0 && !bias_small) return SYNTHSR_EINVAL;
    if (p->bias_shape[c][0] > 0 && (p->bias_shape[c][1] <= 0 || p->bias_shape[c][2] <= 0)) return SYNTHSR_EINVAL;
  }
  const int64_t n = (int64_t)p->out_shape[0] * p->out_shape[1] * p->out_shape[2];
  hipLaunchKernelGGL(deform_gmm_kernel, dim3(syn_grid(n, 256, 256 * 8)), dim3(256), 0, (hipStream_t)stream, labels,
                     field_half, gmm_lut, p->swap_lut_size > 0 ? swap_lut : nullptr, noise, bias_small, seg_out,
                     chan_out, minmax, real_in, real_out, real_minmax, *p);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_mimic_acquisition(const float* in, float* out, const int in_shape[3], const int out_shape[3],
                              const float down_zoom[3], const float up_zoom[3], const float subsample_res[3],
                              int out_stride, int out_offset, int dist_offset, synthsr_stream_t stream) {
  if (!in || !out || bad_shape(in_shape) || bad_shape(out_shape) || !down_zoom || !up_zoom || !subsample_res ||
      out_stride < 1 || out_offset >= out_stride || dist_offset >= out_stride || (out_offset < 0 && dist_offset < 0) ||
      (out_offset >= 0 && out_offset == dist_offset))
    return SYNTHSR_EINVAL;
  MimicP p;
  for (int d = 0; d < 3; ++d) {
    if (!(down_zoom[d] > 0.f) || !(up_zoom[d] > 0.f)) return SYNTHSR_EINVAL;
    p.S[d] = in_shape[d];
    p.R[d] = out_shape[d];
    p.dz[d] = down_zoom[d];
    p.uz[d] = up_zoom[d];
    p.sub[d] = subsample_res[d];
  }
  const int64_t n = (int64_t)out_shape[0] * out_shape[1] * out_shape[2];
  hipLaunchKernelGGL(mimic_acquisition_kernel, dim3(syn_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, p,
                     out_stride, out_offset, dist_offset);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_minmax_init(uint32_t* minmax, int n_pairs, synthsr_stream_t stream) {
  if (!minmax || n_pairs < 1) return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(minmax_init_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, (hipStream_t)stream, minmax, n_pairs);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_minmax_reduce(const float* x, int64_t n, uint32_t* minmax, synthsr_stream_t stream) {
  if (!x || !minmax || n < 1) return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(minmax_reduce_kernel, dim3(syn_grid(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, x, n,
                     minmax);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_normalise_gamma(const float* x, float* out, int64_t n, const uint32_t* minmax, float gexp,
                            synthsr_stream_t stream) {
  if (!x || !out || !minmax || n < 1) return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(normalise_gamma_kernel, dim3(syn_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out, n,
                     minmax, gexp);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_blur3d(const float* in, float* out, const int shape[3], const float* kernel, const int ksize[3],
                   int out_stride, int out_offset, int fill_offset, float fill_value, synthsr_stream_t stream) {
  if (!in || !out || !kernel || bad_shape(shape) || bad_shape(ksize)) return SYNTHSR_EINVAL;
  if (ksize[0] * ksize[1] * ksize[2] > 512 || !(ksize[0] & 1) || !(ksize[1] & 1) || !(ksize[2] & 1))
    return SYNTHSR_EINVAL;
  if (out_stride < 1 || out_offset < 0 || out_offset >= out_stride || fill_offset >= out_stride) return SYNTHSR_EINVAL;
  Shape3 s{{shape[0], shape[1], shape[2]}}, ks{{ksize[0], ksize[1], ksize[2]}};
  const int64_t n = (int64_t)s.d[0] * s.d[1] * s.d[2];
  hipLaunchKernelGGL(blur3d_kernel, dim3(syn_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, s, kernel, ks,
                     out_stride, out_offset, fill_offset, fill_value);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_normalise_blur2(const float* x, const int shape[3], const uint32_t* minmax, float gexp, const float* kernel1,
                            const float* kernel2, float* target, float* image, int image_stride, int image_offset,
                            int fill_offset, float fill_value, synthsr_stream_t stream) {
  if (!x || !minmax || !kernel1 || !kernel2 || !target || !image || bad_shape(shape)) return SYNTHSR_EINVAL;
  if (image_stride < 1 || image_offset < 0 || image_offset >= image_stride || fill_offset >= image_stride ||
      fill_offset == image_offset || x == target || x == image)
    return SYNTHSR_EINVAL;
  Shape3 s{{shape[0], shape[1], shape[2]}};
  const int t0 = (shape[0] + NB_TZ - 1) / NB_TZ, t1 = (shape[1] + NB_TY - 1) / NB_TY, t2 = (shape[2] + NB_TX - 1) / NB_TX;
  const int64_t tiles = (int64_t)t0 * t1 * t2;
  if (tiles >= (1ll << 31)) return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(normalise_blur2_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, x, s, minmax, gexp,
                     kernel1, kernel2, target, image, image_stride, image_offset, fill_offset, fill_value, t1, t2);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_outer3(const float* w, float* out, const int shape[3], int out_stride, int out_offset,
                   synthsr_stream_t stream) {
  if (!w || !out || bad_shape(shape) || out_stride < 1 || out_offset < 0 || out_offset >= out_stride)
    return SYNTHSR_EINVAL;
  Shape3 s{{shape[0], shape[1], shape[2]}};
  const int64_t n = (int64_t)s.d[0] * s.d[1] * s.d[2];
  hipLaunchKernelGGL(outer3_kernel, dim3(syn_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, w, out, s, out_stride,
                     out_offset);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_copy_strided(const float* in, float* out, int64_t n, int in_stride, int in_offset, int out_stride,
                         int out_offset, synthsr_stream_t stream) {
  if (!in || !out || n < 1 || in_stride < 1 || out_stride < 1) return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(copy_strided_kernel, dim3(syn_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, n,
                     in_stride, in_offset, out_stride, out_offset);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_abi_version(void) { return 2; }  // 2: conv context first + caller workspace (include/synthsr_hip.h)
const char* synthsr_build_arch(void) { return "gfx950"; }

}  // extern "C"
