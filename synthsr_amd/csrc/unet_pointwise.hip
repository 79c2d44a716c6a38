// HBM-bound U-Net kernels (gfx950): ELU backward, BatchNorm (stats / apply / backward), fused
// BN+max-pool, fused BN+upsample+concat, 1x1x1 head + L1 loss, Keras-semantics Adam.
// NDHWC: a tensor is [nvox][C]; all kernels move float4 per lane (16 B x 64 lanes = 1 KiB per wave op).
//
// Channel reductions use 384-thread workgroups: every channel count of the U-Net (24..576, all
// multiples of 24) divided by 4 divides 96, so with a grid stride that is a multiple of 384 float4
// lanes each thread keeps a fixed group of 4 channels -> register partials, one LDS pass, one
// global atomic per (block, channel).
#include "common.h"

SYN_DET_SETTER(pointwise)

// workgroups of the channel-reduction kernels: every workgroup ends with one atomicAdd per channel sum on the SAME addresses,
// and same-address atomics serialise at the memory-side unit (~14 ns each): with 2048 workgroups the tail cost 0.6 ms per fp32
// step (29.36 -> 28.76 ms at 768 = 3 per CU; 512 and 1024 are within noise of it).  The grid-stride loops do the rest.
// The head kernels (one atomic per workgroup on the loss word) are insensitive between 256 and 1024.
static constexpr int red_grid() { return 768; }
static constexpr int head_grid() { return 1024; }

namespace {

constexpr int RB = 384;  // reduction block size (6 waves)

struct Shape3 {
  int d[3];
};

__device__ __forceinline__ float elu_grad_from_y(float y) { return y > 0.f ? 1.f : y + 1.f; }  // alpha = 1

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// bf16 activations (BASELINE configs[3] / [4]): the same kernels instantiated on bf16_t move 4 values = 8 bytes per lane and
// do all arithmetic (BatchNorm statistics, reductions, losses) in fp32; stores round to nearest even
struct bf16_t {
  uint16_t v;
};
__device__ __forceinline__ float bf2f(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(bf2f(u.x & 0xffffu), bf2f(u.x >> 16), bf2f(u.y & 0xffffu), bf2f(u.y >> 16));
}
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
  uint2 u;
  u.x = f2bf(v.x) | (f2bf(v.y) << 16);
  u.y = f2bf(v.z) | (f2bf(v.w) << 16);
  *reinterpret_cast<uint2*>(p) = u;
}

// block-level channel reduction of NV float4 partials held by threads with a fixed channel group
template <int NV>
__device__ __forceinline__ void block_channel_reduce(float4 (&part)[NV], int c4, int C4, bool fixed, float* smem) {
  // smem: NV * C4 * 4 floats, zeroed by the caller before accumulation started
  if (fixed) {
    if (syn_det_on()) {
      // deterministic mode: the blockDim / C4 threads of a channel group add one after the other (LDS float atomics land
      // in arbitration order)
      const int rounds = (int)blockDim.x / C4, mine = (int)threadIdx.x / C4;
      for (int r = 0; r < rounds; ++r) {
        if (mine == r) {
#pragma unroll
          for (int k = 0; k < NV; ++k) {
            smem[(k * C4 + c4) * 4 + 0] += part[k].x;
            smem[(k * C4 + c4) * 4 + 1] += part[k].y;
            smem[(k * C4 + c4) * 4 + 2] += part[k].z;
            smem[(k * C4 + c4) * 4 + 3] += part[k].w;
          }
        }
        __syncthreads();
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      atomicAdd(&smem[(k * C4 + c4) * 4 + 0], part[k].x);
      atomicAdd(&smem[(k * C4 + c4) * 4 + 1], part[k].y);
      atomicAdd(&smem[(k * C4 + c4) * 4 + 2], part[k].z);
      atomicAdd(&smem[(k * C4 + c4) * 4 + 3], part[k].w);
    }
  }
  __syncthreads();
}

// one float per thread summed onto *slot (LDS; zeroed and synchronised by the caller).  Deterministic mode: wave butterflies,
// then the waves in order.  Workgroup-uniform call sites only.
__device__ __forceinline__ void block_scalar_add(float v, float* slot) {
  if (!syn_det_on()) {
    atomicAdd(slot, v);
    return;
  }
  __shared__ float wave_part[16];
  v = syn_wave_sum(v);
  if ((threadIdx.x & 63) == 0) wave_part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = *slot;
    for (int w = 0; w < (int)(blockDim.x + 63) / 64; ++w) t += wave_part[w];
    *slot = t;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------ ELU backward
// dz = (dy [+ dy2]) * elu'(y); dbias[c] += sum dz
// Optional fused BatchNorm backward (bn_sums != nullptr): the incoming gradient is w.r.t. the BN OUTPUT of y and is
// first mapped to the BN input, g <- gamma*invstd*(g - sum_dy/n - xhat*sum_dyxhat/n), saving one full pass.
template <typename T>
__global__ __launch_bounds__(RB) void elu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ dy2,
                                                     const T* __restrict__ y, T* __restrict__ dz,
                                                     float* __restrict__ dbias, int64_t n4, int C4,
                                                     const float* __restrict__ bn_stats,
                                                     const float* __restrict__ bn_gamma,
                                                     const float* __restrict__ bn_sums, float eps, float inv_n,
                                                     const float* __restrict__ g1, const float* __restrict__ w1,
                                                     const float* __restrict__ drop = nullptr, int64_t n4ps = 0) {
  // drop != nullptr (feature-wise dropout with one mask per SAMPLE, batchsize > 1: KL.Dropout(noise_shape=[None,1,1,1,C]),
  // ext/neuron/models.py:320-324): y is the conv + ELU output, what followed it (the BatchNorm, or the next conv) read
  // d = s * y with s = drop[sample][c] (n4ps float4 per sample).  The incoming gradient is w.r.t. d (BatchNorm: w.r.t.
  // BN(d), xhat from s * y) and is multiplied by s; dy2 (the skip connection reads y itself) is added unscaled.
  extern __shared__ float smem[];
  const bool fixed = (RB % C4) == 0;
  if (dbias)
    for (int i = threadIdx.x; i < C4 * 4; i += RB) smem[i] = 0.f;
  __syncthreads();
  float4 part[1] = {make_float4(0.f, 0.f, 0.f, 0.f)};
  const int64_t stride = (int64_t)gridDim.x * RB;
  for (int64_t i = blockIdx.x * (int64_t)RB + threadIdx.x; i < n4; i += stride) {
    float4 g;
    if (g1) {  // rank-1 gradient of the 1x1x1 head: dy[v][c] = g1[v] * w1[c], never materialised
      const float gv = g1[i / C4];
      const int c = (int)(i % C4) * 4;
      g = make_float4(gv * w1[c], gv * w1[c + 1], gv * w1[c + 2], gv * w1[c + 3]);
    } else {
      g = ld4(dy + i * 4);
    }
    const float4 a = ld4(y + i * 4);
    float4 sd = make_float4(1.f, 1.f, 1.f, 1.f);
    if (drop) sd = *reinterpret_cast<const float4*>(drop + (i / n4ps) * (C4 * 4) + (i % C4) * 4);
    if (bn_sums) {
      const int C = C4 * 4, c = (int)(i % C4) * 4;
      float gg[4] = {g.x, g.y, g.z, g.w};
      const float aa[4] = {a.x * sd.x, a.y * sd.y, a.z * sd.z, a.w * sd.w};  // the BatchNorm's input (sd = 1 without dropout)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float inv = rsqrtf(bn_stats[C + c + k] + eps);
        const float xh = (aa[k] - bn_stats[c + k]) * inv;
        gg[k] = bn_gamma[c + k] * inv * (gg[k] - bn_sums[c + k] * inv_n - xh * bn_sums[C + c + k] * inv_n);
      }
      g = make_float4(gg[0], gg[1], gg[2], gg[3]);
    }
    if (drop) {
      g.x *= sd.x; g.y *= sd.y; g.z *= sd.z; g.w *= sd.w;
    }
    if (dy2) {
      const float4 g2 = ld4(dy2 + i * 4);
      g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
    }
    float4 r;
    r.x = g.x * elu_grad_from_y(a.x);
    r.y = g.y * elu_grad_from_y(a.y);
    r.z = g.z * elu_grad_from_y(a.z);
    r.w = g.w * elu_grad_from_y(a.w);
    st4(dz + i * 4, r);
    if (dbias) {
      if (fixed) {
        part[0].x += r.x; part[0].y += r.y; part[0].z += r.z; part[0].w += r.w;
      } else {
        const int c4 = (int)(i % C4);
        atomicAdd(&smem[c4 * 4 + 0], r.x);
        atomicAdd(&smem[c4 * 4 + 1], r.y);
        atomicAdd(&smem[c4 * 4 + 2], r.z);
        atomicAdd(&smem[c4 * 4 + 3], r.w);
      }
    }
  }
  if (dbias) {
    block_channel_reduce<1>(part, threadIdx.x % C4, C4, fixed, smem);
    if (syn_det_gather(smem, C4 * 4))
      for (int i = threadIdx.x; i < C4 * 4; i += RB) atomicAdd(&dbias[i], smem[i]);
    syn_det_gather_end(C4 * 4);
  }
}

// out[v][c] = x[v][c] * scale[sample(v)][c]: the dropped-out tensor of a batch with one feature mask per sample (and the same
// factor on a gradient); in place allowed
template <typename T>
__global__ __launch_bounds__(256) void scale_channels_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t n4,
                                                             int C4, const float* __restrict__ scale, int64_t n4ps) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = ld4(x + i * 4);
    const float4 sc = *reinterpret_cast<const float4*>(scale + (i / n4ps) * (C4 * 4) + (i % C4) * 4);
    st4(out + i * 4, make_float4(a.x * sc.x, a.y * sc.y, a.z * sc.z, a.w * sc.w));
  }
}

// ------------------------------------------------------------------------------------------ BN statistics
template <typename T>
__global__ __launch_bounds__(RB) void bn_stats_kernel(const T* __restrict__ x, int64_t n4, int C4,
                                                      double* __restrict__ ws) {
  extern __shared__ float smem[];  // [2][C4*4]
  const bool fixed = (RB % C4) == 0;
  for (int i = threadIdx.x; i < 2 * C4 * 4; i += RB) smem[i] = 0.f;
  __syncthreads();
  float4 part[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  const int64_t stride = (int64_t)gridDim.x * RB;
  for (int64_t i = blockIdx.x * (int64_t)RB + threadIdx.x; i < n4; i += stride) {
    const float4 a = ld4(x + i * 4);
    if (fixed) {
      part[0].x += a.x; part[0].y += a.y; part[0].z += a.z; part[0].w += a.w;
      part[1].x += a.x * a.x; part[1].y += a.y * a.y; part[1].z += a.z * a.z; part[1].w += a.w * a.w;
    } else {
      const int c4 = (int)(i % C4);
      atomicAdd(&smem[c4 * 4 + 0], a.x);
      atomicAdd(&smem[c4 * 4 + 1], a.y);
      atomicAdd(&smem[c4 * 4 + 2], a.z);
      atomicAdd(&smem[c4 * 4 + 3], a.w);
      atomicAdd(&smem[(C4 + c4) * 4 + 0], a.x * a.x);
      atomicAdd(&smem[(C4 + c4) * 4 + 1], a.y * a.y);
      atomicAdd(&smem[(C4 + c4) * 4 + 2], a.z * a.z);
      atomicAdd(&smem[(C4 + c4) * 4 + 3], a.w * a.w);
    }
  }
  block_channel_reduce<2>(part, threadIdx.x % C4, C4, fixed, smem);
  if (syn_det_gather(smem, 2 * C4 * 4))
    for (int i = threadIdx.x; i < 2 * C4 * 4; i += RB) atomicAdd(&ws[i], (double)smem[i]);
  syn_det_gather_end(2 * C4 * 4);
}

__global__ void bn_stats_finalize_kernel(const double* __restrict__ ws, float* __restrict__ stats, int C, double inv_n) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const double m = ws[c] * inv_n;
    double v = ws[C + c] * inv_n - m * m;
    if (v < 0.0) v = 0.0;
    stats[c] = (float)m;
    stats[C + c] = (float)v;
  }
}

// per-channel affine of BN: y = x*sc + sh, sc = gamma*rsqrt(var+eps), sh = beta - mean*sc
__device__ __forceinline__ void bn_coeff(const float* stats, const float* gamma, const float* beta, float eps, int C,
                                         int c, float& sc, float& sh) {
  const float inv = rsqrtf(stats[C + c] + eps) * gamma[c];
  sc = inv;
  sh = beta[c] - stats[c] * inv;
}

__device__ __forceinline__ void bn_coeff4(const float* stats, const float* gamma, const float* beta, float eps, int C,
                                          int c, float4& sc, float4& sh) {
  bn_coeff(stats, gamma, beta, eps, C, c + 0, sc.x, sh.x);
  bn_coeff(stats, gamma, beta, eps, C, c + 1, sc.y, sh.y);
  bn_coeff(stats, gamma, beta, eps, C, c + 2, sc.z, sh.z);
  bn_coeff(stats, gamma, beta, eps, C, c + 3, sc.w, sh.w);
}

__device__ __forceinline__ float4 fma4(float4 a, float4 s, float4 h) {
  return make_float4(a.x * s.x + h.x, a.y * s.y + h.y, a.z * s.z + h.z, a.w * s.w + h.w);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n4,
                                                       int C, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps) {
  const int C4 = C / 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    float4 sc, sh;
    bn_coeff4(stats, gamma, beta, eps, C, c, sc, sh);
    st4(y + i * 4, fma4(ld4(x + i * 4), sc, sh));
  }
}

// ------------------------------------------------------------------------------------------ BN + max-pool 2^3
template <typename T>
__global__ __launch_bounds__(256) void bn_maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, Shape3 s,
                                                         int C, const float* __restrict__ stats,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps) {
  const int C4 = C / 4;
  const int o0 = s.d[0] / 2, o1 = s.d[1] / 2, o2 = s.d[2] / 2;
  const int64_t n4 = (int64_t)o0 * o1 * o2 * C4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    int64_t v = i / C4;
    const int p2 = (int)(v % o2);
    v /= o2;
    const int p1 = (int)(v % o1);
    const int p0 = (int)(v / o1);
    float4 sc, sh;
    bn_coeff4(stats, gamma, beta, eps, C, c, sc, sh);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int64_t vi = ((int64_t)(2 * p0 + a) * s.d[1] + (2 * p1 + b)) * s.d[2] + (2 * p2 + d);
          const float4 t = fma4(ld4(x + vi * C + c), sc, sh);
          m.x = fmaxf(m.x, t.x); m.y = fmaxf(m.y, t.y); m.z = fmaxf(m.z, t.z); m.w = fmaxf(m.w, t.w);
        }
    st4(y + i * 4, m);
  }
}

// gradient w.r.t. the BN output: routed to the FIRST maximum of each 2^3 window (raster order)
template <typename T>
__global__ __launch_bounds__(256) void bn_maxpool_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                             T* __restrict__ dbn, Shape3 s, int C,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps) {
  const int C4 = C / 4;
  const int o0 = s.d[0] / 2, o1 = s.d[1] / 2, o2 = s.d[2] / 2;
  const int64_t n4 = (int64_t)o0 * o1 * o2 * C4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    int64_t v = i / C4;
    const int p2 = (int)(v % o2);
    v /= o2;
    const int p1 = (int)(v % o1);
    const int p0 = (int)(v / o1);
    float4 sc, sh;
    bn_coeff4(stats, gamma, beta, eps, C, c, sc, sh);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int ax = 0, ay = 0, az = 0, aw = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t vi = ((int64_t)(2 * p0 + (k >> 2)) * s.d[1] + (2 * p1 + ((k >> 1) & 1))) * s.d[2] + (2 * p2 + (k & 1));
      const float4 t = fma4(ld4(x + vi * C + c), sc, sh);
      if (t.x > m.x) { m.x = t.x; ax = k; }
      if (t.y > m.y) { m.y = t.y; ay = k; }
      if (t.z > m.z) { m.z = t.z; az = k; }
      if (t.w > m.w) { m.w = t.w; aw = k; }
    }
    const float4 g = ld4(dy + i * 4);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t vi = ((int64_t)(2 * p0 + (k >> 2)) * s.d[1] + (2 * p1 + ((k >> 1) & 1))) * s.d[2] + (2 * p2 + (k & 1));
      st4(dbn + vi * C + c, make_float4(ax == k ? g.x : 0.f, ay == k ? g.y : 0.f, az == k ? g.z : 0.f, aw == k ? g.w : 0.f));
    }
  }
}

// bn_maxpool_bwd + the channel sums of the NEXT BatchNorm backward (synthsr_bn_bwd_reduce of the routed gradient):
// the routed gradient is 7/8 zeros, every non-zero and its xhat are in registers here, so the separate reduction pass
// (a full read of dbn and x) is unnecessary.  RB threads so that each thread keeps a fixed channel group.
template <typename T>
__global__ __launch_bounds__(RB) void bn_maxpool_bwd_sums_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                 T* __restrict__ dbn, Shape3 s, int C,
                                                                 const float* __restrict__ stats,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float eps,
                                                                 float* __restrict__ sums) {
  extern __shared__ float smem[];  // [2][C]
  const int C4 = C / 4;
  const bool fixed = (RB % C4) == 0;
  for (int i = threadIdx.x; i < 2 * C; i += RB) smem[i] = 0.f;
  __syncthreads();
  float4 part[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  const int o0 = s.d[0] / 2, o1 = s.d[1] / 2, o2 = s.d[2] / 2;
  const int64_t n4 = (int64_t)o0 * o1 * o2 * C4;
  for (int64_t i = blockIdx.x * (int64_t)RB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * RB) {
    const int c = (int)(i % C4) * 4;
    int64_t v = i / C4;
    const int p2 = (int)(v % o2);
    v /= o2;
    const int p1 = (int)(v % o1);
    const int p0 = (int)(v / o1);
    float4 sc, sh;
    bn_coeff4(stats, gamma, beta, eps, C, c, sc, sh);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    float4 xa = make_float4(0.f, 0.f, 0.f, 0.f);  // raw input at the arg-max
    int ax = 0, ay = 0, az = 0, aw = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t vi = ((int64_t)(2 * p0 + (k >> 2)) * s.d[1] + (2 * p1 + ((k >> 1) & 1))) * s.d[2] + (2 * p2 + (k & 1));
      const float4 r = ld4(x + vi * C + c);
      const float4 t = fma4(r, sc, sh);
      if (t.x > m.x) { m.x = t.x; ax = k; xa.x = r.x; }
      if (t.y > m.y) { m.y = t.y; ay = k; xa.y = r.y; }
      if (t.z > m.z) { m.z = t.z; az = k; xa.z = r.z; }
      if (t.w > m.w) { m.w = t.w; aw = k; xa.w = r.w; }
    }
    const float4 g = ld4(dy + i * 4);
    if (dbn) {  // nullptr: the sums only -- the routed gradient is re-derived by bn_pool_elu_bwd_kernel
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int64_t vi = ((int64_t)(2 * p0 + (k >> 2)) * s.d[1] + (2 * p1 + ((k >> 1) & 1))) * s.d[2] + (2 * p2 + (k & 1));
        st4(dbn + vi * C + c, make_float4(ax == k ? g.x : 0.f, ay == k ? g.y : 0.f, az == k ? g.z : 0.f, aw == k ? g.w : 0.f));
      }
    }
    float4 gx;  // g * xhat(arg-max)
    gx.x = g.x * (xa.x - stats[c + 0]) * rsqrtf(stats[C + c + 0] + eps);
    gx.y = g.y * (xa.y - stats[c + 1]) * rsqrtf(stats[C + c + 1] + eps);
    gx.z = g.z * (xa.z - stats[c + 2]) * rsqrtf(stats[C + c + 2] + eps);
    gx.w = g.w * (xa.w - stats[c + 3]) * rsqrtf(stats[C + c + 3] + eps);
    if (fixed) {
      part[0].x += g.x; part[0].y += g.y; part[0].z += g.z; part[0].w += g.w;
      part[1].x += gx.x; part[1].y += gx.y; part[1].z += gx.z; part[1].w += gx.w;
    } else {
      atomicAdd(&smem[c + 0], g.x); atomicAdd(&smem[c + 1], g.y);
      atomicAdd(&smem[c + 2], g.z); atomicAdd(&smem[c + 3], g.w);
      atomicAdd(&smem[C + c + 0], gx.x); atomicAdd(&smem[C + c + 1], gx.y);
      atomicAdd(&smem[C + c + 2], gx.z); atomicAdd(&smem[C + c + 3], gx.w);
    }
  }
  block_channel_reduce<2>(part, threadIdx.x % C4, C4, fixed, smem);
  if (syn_det_gather(smem, 2 * C))
    for (int i = threadIdx.x; i < 2 * C; i += RB) atomicAdd(&sums[i], smem[i]);
  syn_det_gather_end(2 * C);
}

// MaxPooling3D backward + BatchNormalization backward (pass 2) + ELU backward of an encoder level in ONE pass
// (ext/neuron/models.py:316-356: conv + ELU -> BatchNormalization -> MaxPooling3D): dz = (BN'(route(dpool)) + dy2) * ELU'(y).
// The gradient routed to the arg-max of every 2x2x2 window is 7/8 zeros; writing it out (bn_maxpool_bwd) and reading it back
// (elu_bwd) cost one write + one read of the level's largest tensor (2 x 393 MB at 160^3 x 24).  Here each thread owns a
// pooling window x 4 channels: it re-derives the arg-max from y (same arithmetic as bn_maxpool_kernel: first maximum in
// raster order of y * sc + sh), applies g <- gamma * invstd * (g - sum_dy / n - xhat * sum_dyxhat / n) to all 8 voxels
// (sums from bn_maxpool_bwd_sums_kernel with dbn = nullptr), adds the skip connection's gradient dy2 and multiplies by
// ELU'(y); dbias += sum dz.
template <typename T>
__global__ __launch_bounds__(RB) void bn_pool_elu_bwd_kernel(const T* __restrict__ dpool, const T* __restrict__ y,
                                                             const T* __restrict__ dy2, T* __restrict__ dz,
                                                             float* __restrict__ dbias, Shape3 s, int C,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ sums, float eps, float inv_n) {
  extern __shared__ float smem[];  // [C]
  const int C4 = C / 4;
  const bool fixed = (RB % C4) == 0;
  if (dbias)
    for (int i = threadIdx.x; i < C; i += RB) smem[i] = 0.f;
  __syncthreads();
  float4 part[1] = {make_float4(0.f, 0.f, 0.f, 0.f)};
  const int o0 = s.d[0] / 2, o1 = s.d[1] / 2, o2 = s.d[2] / 2;
  const int64_t n4 = (int64_t)o0 * o1 * o2 * C4;
  for (int64_t i = blockIdx.x * (int64_t)RB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * RB) {
    const int c = (int)(i % C4) * 4;
    int64_t v = i / C4;
    const int p2 = (int)(v % o2);
    v /= o2;
    const int p1 = (int)(v % o1);
    const int p0 = (int)(v / o1);
    float4 sc, sh;
    bn_coeff4(stats, gamma, beta, eps, C, c, sc, sh);
    float mean[4], inv[4], gam[4], s0[4], s1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mean[k] = stats[c + k];
      inv[k] = rsqrtf(stats[C + c + k] + eps);
      gam[k] = gamma[c + k];
      s0[k] = sums[c + k];
      s1[k] = sums[C + c + k];
    }
    float4 r[8];
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int ax = 0, ay = 0, az = 0, aw = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t vi = ((int64_t)(2 * p0 + (k >> 2)) * s.d[1] + (2 * p1 + ((k >> 1) & 1))) * s.d[2] + (2 * p2 + (k & 1));
      r[k] = ld4(y + vi * C + c);
      const float4 t = fma4(r[k], sc, sh);
      if (t.x > m.x) { m.x = t.x; ax = k; }
      if (t.y > m.y) { m.y = t.y; ay = k; }
      if (t.z > m.z) { m.z = t.z; az = k; }
      if (t.w > m.w) { m.w = t.w; aw = k; }
    }
    const float4 g = ld4(dpool + i * 4);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t vi = ((int64_t)(2 * p0 + (k >> 2)) * s.d[1] + (2 * p1 + ((k >> 1) & 1))) * s.d[2] + (2 * p2 + (k & 1));
      float gg[4] = {ax == k ? g.x : 0.f, ay == k ? g.y : 0.f, az == k ? g.z : 0.f, aw == k ? g.w : 0.f};
      const float aa[4] = {r[k].x, r[k].y, r[k].z, r[k].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xh = (aa[q] - mean[q]) * inv[q];
        gg[q] = gam[q] * inv[q] * (gg[q] - s0[q] * inv_n - xh * s1[q] * inv_n);  // the expression of elu_bwd_kernel, bit for bit
      }
      if (dy2) {
        const float4 g2 = ld4(dy2 + vi * C + c);
        gg[0] += g2.x; gg[1] += g2.y; gg[2] += g2.z; gg[3] += g2.w;
      }
      float4 o;
      o.x = gg[0] * elu_grad_from_y(aa[0]);
      o.y = gg[1] * elu_grad_from_y(aa[1]);
      o.z = gg[2] * elu_grad_from_y(aa[2]);
      o.w = gg[3] * elu_grad_from_y(aa[3]);
      st4(dz + vi * C + c, o);
      if (dbias) {
        if (fixed) {
          part[0].x += o.x; part[0].y += o.y; part[0].z += o.z; part[0].w += o.w;
        } else {
          atomicAdd(&smem[c + 0], o.x);
          atomicAdd(&smem[c + 1], o.y);
          atomicAdd(&smem[c + 2], o.z);
          atomicAdd(&smem[c + 3], o.w);
        }
      }
    }
  }
  if (dbias) {
    block_channel_reduce<1>(part, threadIdx.x % C4, C4, fixed, smem);
    if (syn_det_gather(smem, C))
      for (int i = threadIdx.x; i < C; i += RB) atomicAdd(&dbias[i], smem[i]);
    syn_det_gather_end(C);
  }
}

// ------------------------------------------------------------------------------------------ BN backward
template <typename T>
__global__ __launch_bounds__(RB) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           int64_t n4, int C, const float* __restrict__ stats,
                                                           float eps, float* __restrict__ sums) {
  extern __shared__ float smem[];  // [2][C]
  const int C4 = C / 4;
  const bool fixed = (RB % C4) == 0;
  for (int i = threadIdx.x; i < 2 * C; i += RB) smem[i] = 0.f;
  __syncthreads();
  float4 part[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  const int64_t stride = (int64_t)gridDim.x * RB;
  for (int64_t i = blockIdx.x * (int64_t)RB + threadIdx.x; i < n4; i += stride) {
    const int c = (int)(i % C4) * 4;
    const float4 g = ld4(dy + i * 4);
    const float4 a = ld4(x + i * 4);
    float4 xh;
    xh.x = (a.x - stats[c + 0]) * rsqrtf(stats[C + c + 0] + eps);
    xh.y = (a.y - stats[c + 1]) * rsqrtf(stats[C + c + 1] + eps);
    xh.z = (a.z - stats[c + 2]) * rsqrtf(stats[C + c + 2] + eps);
    xh.w = (a.w - stats[c + 3]) * rsqrtf(stats[C + c + 3] + eps);
    if (fixed) {
      part[0].x += g.x; part[0].y += g.y; part[0].z += g.z; part[0].w += g.w;
      part[1].x += g.x * xh.x; part[1].y += g.y * xh.y; part[1].z += g.z * xh.z; part[1].w += g.w * xh.w;
    } else {
      atomicAdd(&smem[c + 0], g.x); atomicAdd(&smem[c + 1], g.y);
      atomicAdd(&smem[c + 2], g.z); atomicAdd(&smem[c + 3], g.w);
      atomicAdd(&smem[C + c + 0], g.x * xh.x); atomicAdd(&smem[C + c + 1], g.y * xh.y);
      atomicAdd(&smem[C + c + 2], g.z * xh.z); atomicAdd(&smem[C + c + 3], g.w * xh.w);
    }
  }
  block_channel_reduce<2>(part, threadIdx.x % C4, C4, fixed, smem);
  if (syn_det_gather(smem, 2 * C))
    for (int i = threadIdx.x; i < 2 * C; i += RB) atomicAdd(&sums[i], smem[i]);
  syn_det_gather_end(2 * C);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           float* __restrict__ dx, int64_t n4, int C,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, float eps,
                                                           const float* __restrict__ sums, float inv_n) {
  const int C4 = C / 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const float4 g = ld4(dy + i * 4);
    const float4 a = ld4(x + i * 4);
    float r[4];
    const float gg[4] = {g.x, g.y, g.z, g.w}, aa[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float inv = rsqrtf(stats[C + c + k] + eps);
      const float xh = (aa[k] - stats[c + k]) * inv;
      r[k] = gamma[c + k] * inv * (gg[k] - sums[c + k] * inv_n - xh * sums[C + c + k] * inv_n);
    }
    st4(dx + i * 4, make_float4(r[0], r[1], r[2], r[3]));
  }
}

// ------------------------------------------------------------------------------------------ upsample + concat
template <typename T>
__global__ __launch_bounds__(256) void upsample_concat_kernel(const T* __restrict__ skip,
                                                              const T* __restrict__ lo, T* __restrict__ out,
                                                              Shape3 s, int Cs, int Cl,
                                                              const float* __restrict__ stats,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps) {
  const int C = Cs + Cl, C4 = C / 4;
  const int64_t n4 = (int64_t)s.d[0] * s.d[1] * s.d[2] * C4;
  const int l1 = s.d[1] / 2, l2 = s.d[2] / 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    int64_t v = i / C4;
    float4 r;
    if (c < Cs) {
      r = ld4(skip + v * Cs + c);
    } else {
      const int i2 = (int)(v % s.d[2]);
      const int i1 = (int)((v / s.d[2]) % s.d[1]);
      const int i0 = (int)(v / ((int64_t)s.d[2] * s.d[1]));
      const int64_t lv = ((int64_t)(i0 >> 1) * l1 + (i1 >> 1)) * l2 + (i2 >> 1);
      float4 sc, sh;
      bn_coeff4(stats, gamma, beta, eps, Cl, c - Cs, sc, sh);
      r = fma4(ld4(lo + lv * Cl + (c - Cs)), sc, sh);
    }
    st4(out + i * 4, r);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample_concat_bwd_kernel(const T* __restrict__ dcat,
                                                                  T* __restrict__ dskip,
                                                                  T* __restrict__ dlo, Shape3 s, int Cs, int Cl) {
  const int C = Cs + Cl;
  const int Cs4 = Cs / 4, Cl4 = Cl / 4;
  const int64_t nv = (int64_t)s.d[0] * s.d[1] * s.d[2];
  const int l0 = s.d[0] / 2, l1 = s.d[1] / 2, l2 = s.d[2] / 2;
  const int64_t nskip4 = nv * Cs4;
  const int64_t nlo4 = (int64_t)l0 * l1 * l2 * Cl4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nskip4 + nlo4;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nskip4) {
      const int c = (int)(i % Cs4) * 4;
      const int64_t v = i / Cs4;
      st4(dskip + v * Cs + c, ld4(dcat + v * C + c));
    } else {
      const int64_t j = i - nskip4;
      const int c = (int)(j % Cl4) * 4;
      int64_t v = j / Cl4;
      const int p2 = (int)(v % l2);
      v /= l2;
      const int p1 = (int)(v % l1);
      const int p0 = (int)(v / l1);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int64_t vi = ((int64_t)(2 * p0 + (k >> 2)) * s.d[1] + (2 * p1 + ((k >> 1) & 1))) * s.d[2] + (2 * p2 + (k & 1));
        const float4 t = ld4(dcat + vi * C + Cs + c);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
      }
      st4(dlo + j * 4, acc);
    }
  }
}

// ------------------------------------------------------------------------------------------ head + regression loss
// unet_likelihood (1x1x1 conv on the last BatchNorm output, K output channels) + the loss of SynthSR/metrics_model.py:30-132
// with n regression targets (training(output_channel=[...]), target [nvox][n]):
//   kind 0  'l1'      mean |pred - target|                          (K = n)
//   kind 1  'l2'      mean (pred - target)^2                        (K = n)
//   kind 2  'laplace' mean( log(2 b) + |pred_k - target_k| / b ),  b = 1e-5 + 0.02 exp(pred_{n+k})     (K = 2 n)
// (means over voxels AND target channels, like K.mean in the reference)
// optionally evaluated on a centred box only (loss_cropping, metrics_model.py:70-90): voxels outside contribute neither
// loss nor gradient; inv_n = 1 / (voxels inside).  dpred [nvox][K] is the loss gradient w.r.t. pred.
struct HeadBox {
  int on, d1, d2;
  int lo[3], hi[3];
  int res_off[4];  // residual channel added to intensity channel k (work_with_residual_channel)
};

// CT = 24 (round 6): the 24-feature head of the benchmark network keeps the 72 BatchNorm / head coefficients of its dot product in
// SCALAR registers (readfirstlane: the values are uniform) instead of reading them from LDS per voxel -- 72 broadcast ds_read_b32
// per voxel and pass -- and requests the next 256-voxel slab before it computes on the current one: 0.184 -> 0.153 ms at 160^3
// (2.1 -> 2.6 TB/s; profiles/r06_head_kernel_ab.txt; the VGPR form of the same idea needs 189 registers and is slower).  Same
// products in the same order; the compiler's fma contraction may differ in the last bit.  CT = 0: any channel count, LDS.
template <typename T, int K, int CT = 0>
__global__ __launch_bounds__(256) void head_loss_fwd_kernel(const T* __restrict__ x, int64_t nvox, int C,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            const float* __restrict__ residual, int rs,
                                                            const float* __restrict__ target, float* __restrict__ pred,
                                                            float* __restrict__ dpred, float* __restrict__ loss,
                                                            float inv_n, int kind, HeadBox box, float* __restrict__ ab) {
  // ab != nullptr (K == 1): also the two sums the backward pass of this head needs, A[c] = sum_v g[v] xhat[v][c] and
  // B = sum_v g[v] with g = d loss / d pred (ab[0 .. C) += A, ab[C] += B): the gradient w.r.t. the BatchNorm output is rank-1
  // (g[v] w[c]), so head-weight gradients and that BatchNorm's backward sums are linear in (A, B) (head_ab_finish_kernel) and
  // the separate pass over the last feature map (head_bwd_kernel, 393 MB at 160^3) is not needed.
  constexpr int NTMAX = K;
  const int NT = kind == 2 ? K / 2 : K;  // regression targets
  extern __shared__ float smem[];  // scale[C], shift[C], w[C][K], then the voxel tile
  float* weff = smem;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float sc, sh;
    bn_coeff(stats, gamma, beta, eps, C, c, sc, sh);
    weff[c] = sc;         // keep scale and shift separately: pred accumulates w*(x*sc+sh) like the unfused graph
    weff[C + c] = sh;
#pragma unroll
    for (int k = 0; k < K; ++k) weff[2 * C + k * C + c] = w[c * K + k];
  }
  __syncthreads();
  constexpr int CR = (CT > 0 && K == 1) ? CT : 1;
  float rsc[CR], rsh[CR], rw[CR];
  if constexpr (CT > 0 && K == 1) {
#pragma unroll
    for (int c = 0; c < CR; ++c) {
      rsc[c] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, weff[c])));       // scalar
      rsh[c] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, weff[C + c])));   // registers
      rw[c] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, weff[2 * C + c])));
    }
  }
  float lsum = 0.f;
  // second phase of a pass (ab): thread = (channel quad qa, voxel lane la); la walks the tile's voxels with stride LA
  __shared__ float gl[256];
  const int C4h = C / 4, LA = 256 / C4h;
  const int qa = (int)threadIdx.x % C4h, la = (int)threadIdx.x / C4h;
  float4 apart[1] = {make_float4(0.f, 0.f, 0.f, 0.f)};
  float bsum = 0.f;
  float amean[4] = {0.f, 0.f, 0.f, 0.f}, ainv[4] = {0.f, 0.f, 0.f, 0.f};
  if (ab) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      amean[k] = stats[qa * 4 + k];
      ainv[k] = rsqrtf(stats[C + qa * 4 + k] + eps);
    }
  }
  // 256 voxels per pass: the [256][C] slab is contiguous in memory -> coalesced float4 loads into an LDS tile with rows
  // padded to C+4 floats (conflict-free 16-byte reads), then one thread per voxel (thread-per-voxel global loads touch 48
  // cache lines per instruction and ran at 1.7 TB/s)
  float* tile = smem + (2 + K) * C;
  const int C4 = C / 4, CP = C + 4;
  // round 6: the slab of the NEXT pass is requested (registers) before this pass computes, so that the loads of a workgroup
  // are in flight during its two compute phases instead of being waited for between three barriers
  constexpr int PF = 8;                 // float4 per thread of a prefetched slab: C <= 32
  const bool prefetch = C4 <= PF;
  float4 nxt[PF];
  auto fetch = [&](int64_t v0n) {
    const int nvn = v0n < nvox ? (int)min((int64_t)256, nvox - v0n) : 0;
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const int i = (int)threadIdx.x + 256 * k;
      if (k < C4 && i < nvn * C4) nxt[k] = ld4(x + v0n * C + (int64_t)i * 4);
    }
  };
  if (prefetch) fetch((int64_t)blockIdx.x * 256);
  for (int64_t v0 = (int64_t)blockIdx.x * 256; v0 < nvox; v0 += (int64_t)gridDim.x * 256) {
    const int nv = (int)min((int64_t)256, nvox - v0);
    __syncthreads();
    if (prefetch) {
#pragma unroll
      for (int k = 0; k < PF; ++k) {
        const int i = (int)threadIdx.x + 256 * k;
        if (k < C4 && i < nv * C4) {
          const int vl = i / C4, q = i - vl * C4;
          *reinterpret_cast<float4*>(&tile[vl * CP + q * 4]) = nxt[k];
        }
      }
      fetch(v0 + (int64_t)gridDim.x * 256);
    } else {
      for (int i = threadIdx.x; i < nv * C4; i += 256) {
        const int vl = i / C4, q = i - vl * C4;
        *reinterpret_cast<float4*>(&tile[vl * CP + q * 4]) = ld4(x + v0 * C + (int64_t)i * 4);
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < nv) {
      const int64_t v = v0 + threadIdx.x;
      float acc[K];
#pragma unroll
      for (int k = 0; k < K; ++k) acc[k] = 0.f;
      const float* xp = tile + threadIdx.x * CP;
      if constexpr (CT > 0 && K == 1) {
#pragma unroll
        for (int c = 0; c < CR; c += 4) {
          const float4 a = *reinterpret_cast<const float4*>(xp + c);
          const float n0 = a.x * rsc[c + 0] + rsh[c + 0], n1 = a.y * rsc[c + 1] + rsh[c + 1];
          const float n2 = a.z * rsc[c + 2] + rsh[c + 2], n3 = a.w * rsc[c + 3] + rsh[c + 3];
          acc[0] += rw[c + 0] * n0;
          acc[0] += rw[c + 1] * n1;
          acc[0] += rw[c + 2] * n2;
          acc[0] += rw[c + 3] * n3;
        }
      } else
      for (int c = 0; c < C; c += 4) {
        const float4 a = *reinterpret_cast<const float4*>(xp + c);
        const float n0 = a.x * weff[c + 0] + weff[C + c + 0], n1 = a.y * weff[c + 1] + weff[C + c + 1];
        const float n2 = a.z * weff[c + 2] + weff[C + c + 2], n3 = a.w * weff[c + 3] + weff[C + c + 3];
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const float* wk = weff + 2 * C + k * C + c;
          acc[k] += wk[0] * n0;
          acc[k] += wk[1] * n1;
          acc[k] += wk[2] * n2;
          acc[k] += wk[3] * n3;
        }
      }
#pragma unroll
      for (int k = 0; k < K; ++k) acc[k] += b[k];
      if (residual) {
#pragma unroll
        for (int k = 0; k < NTMAX; ++k)
          if (k < NT) acc[k] += residual[v * rs + box.res_off[k]];
      }
      if (pred) {
#pragma unroll
        for (int k = 0; k < K; ++k) pred[v * K + k] = acc[k];
      }
      bool inside = true;
      if (box.on) {
        const int xx = (int)(v % box.d2), yy = (int)((v / box.d2) % box.d1), zz = (int)(v / ((int64_t)box.d1 * box.d2));
        inside = zz >= box.lo[0] && zz < box.hi[0] && yy >= box.lo[1] && yy < box.hi[1] && xx >= box.lo[2] && xx < box.hi[2];
      }
      float g[K];
#pragma unroll
      for (int k = 0; k < K; ++k) g[k] = 0.f;
      if (inside) {
#pragma unroll
        for (int k = 0; k < NTMAX; ++k) {
          if (k >= NT) continue;
          const float e = acc[k] - target[v * NT + k];
          const float sgn = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
          if (kind == 2) {  // laplace: spread channel NT + k
            if constexpr (K >= 2) {
              const int ks = (K / 2) + k;  // == NT + k
              const float ex = 0.02f * expf(acc[ks < K ? ks : 0]), bb = 1e-5f + ex, ib = 1.f / bb;
              lsum += logf(2.f * bb) + fabsf(e) * ib;
              g[k] = sgn * ib * inv_n;
              g[ks < K ? ks : 0] = (ib - fabsf(e) * ib * ib) * ex * inv_n;
            }
          } else if (kind == 1) {
            lsum += e * e;
            g[k] = 2.f * e * inv_n;
          } else {
            lsum += fabsf(e);
            g[k] = sgn * inv_n;
          }
        }
      }
      if (dpred) {
#pragma unroll
        for (int k = 0; k < K; ++k) dpred[v * K + k] = g[k];
      }
      if (ab) {
        gl[threadIdx.x] = g[0];
        bsum += g[0];
      }
    }
    if (ab) {
      __syncthreads();
      if (la < LA) {
        for (int vl = la; vl < nv; vl += LA) {
          const float gv = gl[vl];
          const float4 a = *reinterpret_cast<const float4*>(tile + vl * CP + qa * 4);
          apart[0].x += gv * ((a.x - amean[0]) * ainv[0]);
          apart[0].y += gv * ((a.y - amean[1]) * ainv[1]);
          apart[0].z += gv * ((a.z - amean[2]) * ainv[2]);
          apart[0].w += gv * ((a.w - amean[3]) * ainv[3]);
        }
      }
    }
  }
  // one atomic per BLOCK on the single loss word: 16k same-address atomics (one per wave of a 4096-block grid) used to
  // serialise into ~0.15 ms
  lsum = syn_wave_sum(lsum);
  __shared__ float wsum[4];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) wsum[0] = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * inv_n;
  __syncthreads();
  if (!ab) {
    if (syn_det_gather(wsum, 1))
      if (threadIdx.x == 0) atomicAdd(loss, wsum[0]);
    syn_det_gather_end(1);
    return;
  }
  // one row [A (C) | B | loss] per workgroup: ONE gather (two in a row would mix their arrival counts)
  __shared__ float row[128];
  for (int i = threadIdx.x; i < C + 2; i += 256) row[i] = 0.f;
  __syncthreads();
  block_scalar_add(bsum, &row[C]);
  if (la >= LA) apart[0] = make_float4(0.f, 0.f, 0.f, 0.f);
  block_channel_reduce<1>(apart, qa, C4h, true, row);
  if (threadIdx.x == 0) row[C + 1] = wsum[0];
  __syncthreads();
  if (syn_det_gather(row, C + 2)) {
    for (int i = threadIdx.x; i < C + 1; i += 256) atomicAdd(&ab[i], row[i]);
    if (threadIdx.x == 0) atomicAdd(loss, row[C + 1]);
  }
  syn_det_gather_end(C + 2);
}

// backward of the 1-channel head from the sums of head_loss_fwd_kernel (see there): dw = gamma A + beta B, db = B, and the
// BatchNorm-backward sums of the rank-1 gradient g[v] w[c]: sum = w B, sum * xhat = w A
__global__ void head_ab_finish_kernel(const float* __restrict__ ab, int C, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, const float* __restrict__ w, float* __restrict__ dw,
                                      float* __restrict__ db, float* __restrict__ sums) {
  const float B = ab[C];
  for (int i = threadIdx.x; i < C; i += blockDim.x) {
    const float A = ab[i];
    dw[i] += gamma[i] * A + beta[i] * B;
    if (sums) {
      sums[i] += w[i] * B;
      sums[C + i] += w[i] * A;
    }
  }
  if (threadIdx.x == 0) *db += B;
}

// K-channel head backward (K > 1, e.g. the intensity + spread channels of the 'laplace' loss): the gradient w.r.t. the
// BatchNorm output, dbn[v][c] = sum_k g[v][k] w[c][k], is written out; dw[c][k] += gamma[c] A_k[c] + beta[c] B_k,
// db[k] += B_k with A_k[c] = sum_v g_k xhat[v][c], B_k = sum_v g_k.
template <typename T, int K>
__global__ __launch_bounds__(RB) void head_multi_bwd_kernel(const float* __restrict__ dpred, const T* __restrict__ x,
                                                            int64_t n4, int C, const float* __restrict__ stats,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            const float* __restrict__ w, T* __restrict__ dbn,
                                                            float* __restrict__ dw, float* __restrict__ db) {
  extern __shared__ float smem[];  // A[K][C], B[K]
  const int C4 = C / 4;
  const bool fixed = (RB % C4) == 0;
  for (int i = threadIdx.x; i < K * C + K; i += RB) smem[i] = 0.f;
  __syncthreads();
  float4 part[K];
  float dbp[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    part[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    dbp[k] = 0.f;
  }
  const int64_t stride = (int64_t)gridDim.x * RB;
  for (int64_t i = blockIdx.x * (int64_t)RB + threadIdx.x; i < n4; i += stride) {
    const int c = (int)(i % C4) * 4;
    const int64_t v = i / C4;
    const float4 a = ld4(x + i * 4);
    float4 xh;
    xh.x = (a.x - stats[c + 0]) * rsqrtf(stats[C + c + 0] + eps);
    xh.y = (a.y - stats[c + 1]) * rsqrtf(stats[C + c + 1] + eps);
    xh.z = (a.z - stats[c + 2]) * rsqrtf(stats[C + c + 2] + eps);
    xh.w = (a.w - stats[c + 3]) * rsqrtf(stats[C + c + 3] + eps);
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float g = dpred[v * K + k];
      d.x += g * w[(c + 0) * K + k]; d.y += g * w[(c + 1) * K + k];
      d.z += g * w[(c + 2) * K + k]; d.w += g * w[(c + 3) * K + k];
      if (fixed) {
        part[k].x += g * xh.x; part[k].y += g * xh.y; part[k].z += g * xh.z; part[k].w += g * xh.w;
      } else {
        atomicAdd(&smem[k * C + c + 0], g * xh.x); atomicAdd(&smem[k * C + c + 1], g * xh.y);
        atomicAdd(&smem[k * C + c + 2], g * xh.z); atomicAdd(&smem[k * C + c + 3], g * xh.w);
      }
      if (c == 0) dbp[k] += g;
    }
    st4(dbn + i * 4, d);
  }
#pragma unroll
  for (int k = 0; k < K; ++k) block_scalar_add(dbp[k], &smem[K * C + k]);
  block_channel_reduce<K>(part, threadIdx.x % C4, C4, fixed, smem);
  if (syn_det_gather(smem, K * C + K)) {
    for (int i = threadIdx.x; i < C * K; i += RB) {
      const int c = i / K, k = i - c * K;
      atomicAdd(&dw[i], gamma[c] * smem[k * C + c] + beta[c] * smem[K * C + k]);
    }
    if (threadIdx.x < K) atomicAdd(&db[threadIdx.x], smem[K * C + threadIdx.x]);
  }
  syn_det_gather_end(K * C + K);
}

template <typename T>
__global__ __launch_bounds__(RB) void head_bwd_kernel(const float* __restrict__ dpred, const T* __restrict__ x,
                                                      int64_t n4, int C, const float* __restrict__ stats,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps,
                                                      const float* __restrict__ w, T* __restrict__ dbn,
                                                      float* __restrict__ dw, float* __restrict__ db,
                                                      float* __restrict__ sums) {
  // The gradient w.r.t. the BatchNorm output is rank-1, dbn[v][c] = g[v] w[c].  With A[c] = sum_v g xhat[v][c] and
  // B = sum_v g everything downstream is linear in (A, B):  dw = gamma A + beta B,  db = B, and the BN-backward sums
  // sum dbn = w B, sum dbn xhat = w A -- so dbn need not be written (dbn == nullptr) nor reduced again (sums != nullptr).
  extern __shared__ float smem[];  // [C] A partials + 1 B
  const int C4 = C / 4;
  const bool fixed = (RB % C4) == 0;
  for (int i = threadIdx.x; i < C + 1; i += RB) smem[i] = 0.f;
  __syncthreads();
  float4 part[1] = {make_float4(0.f, 0.f, 0.f, 0.f)};
  float dbp = 0.f;
  const int64_t stride = (int64_t)gridDim.x * RB;
  for (int64_t i = blockIdx.x * (int64_t)RB + threadIdx.x; i < n4; i += stride) {
    const int c = (int)(i % C4) * 4;
    const int64_t v = i / C4;
    const float g = dpred[v];
    const float4 a = ld4(x + i * 4);
    float4 xh;
    xh.x = (a.x - stats[c + 0]) * rsqrtf(stats[C + c + 0] + eps);
    xh.y = (a.y - stats[c + 1]) * rsqrtf(stats[C + c + 1] + eps);
    xh.z = (a.z - stats[c + 2]) * rsqrtf(stats[C + c + 2] + eps);
    xh.w = (a.w - stats[c + 3]) * rsqrtf(stats[C + c + 3] + eps);
    if (dbn) st4(dbn + i * 4, make_float4(g * w[c + 0], g * w[c + 1], g * w[c + 2], g * w[c + 3]));
    if (fixed) {
      part[0].x += g * xh.x; part[0].y += g * xh.y; part[0].z += g * xh.z; part[0].w += g * xh.w;
    } else {
      atomicAdd(&smem[c + 0], g * xh.x); atomicAdd(&smem[c + 1], g * xh.y);
      atomicAdd(&smem[c + 2], g * xh.z); atomicAdd(&smem[c + 3], g * xh.w);
    }
    if (c == 0) dbp += g;
  }
  block_scalar_add(dbp, &smem[C]);
  block_channel_reduce<1>(part, threadIdx.x % C4, C4, fixed, smem);
  if (syn_det_gather(smem, C + 1)) {
    const float B = smem[C];
    for (int i = threadIdx.x; i < C; i += RB) {
      const float A = smem[i];
      atomicAdd(&dw[i], gamma[i] * A + beta[i] * B);
      if (sums) {
        atomicAdd(&sums[i], w[i] * B);
        atomicAdd(&sums[C + i], w[i] * A);
      }
    }
    if (threadIdx.x == 0) atomicAdd(db, B);
  }
  syn_det_gather_end(C + 1);
}

// ------------------------------------------------------------------ segmentation-regularised loss (metrics_model.py:136-215)
// Frozen segmentation U-Net head: probs[v][n] = softmax_n( sum_c W[c][n] * bn(x[v][c]) + b[n] )  (unet_likelihood, 1x1x1 conv +
// softmax, models.py:480-494).  One thread per voxel, weights in LDS.  C <= 64, N <= 64.
constexpr int SEG_MAXC = 64, SEG_MAXN = 64, SEG_MAXK = 64;

__global__ __launch_bounds__(256) void seg_head_fwd_kernel(const float* __restrict__ x, int64_t nvox, int C,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps,
                                                           const float* __restrict__ W, const float* __restrict__ b, int N,
                                                           float* __restrict__ probs) {
  __shared__ float sw[SEG_MAXC * SEG_MAXN], ssc[SEG_MAXC], ssh[SEG_MAXC], sb[SEG_MAXN];
  for (int i = threadIdx.x; i < C * N; i += blockDim.x) sw[i] = W[i];
  for (int i = threadIdx.x; i < C; i += blockDim.x) bn_coeff(stats, gamma, beta, eps, C, i, ssc[i], ssh[i]);
  for (int i = threadIdx.x; i < N; i += blockDim.x) sb[i] = b[i];
  __syncthreads();
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * blockDim.x) {
    float* pr = probs + v * N;
    float mx = -INFINITY;
    for (int n = 0; n < N; ++n) {  // logits parked in the output row (L1/L2 resident), then normalised in place
      float acc = sb[n];
      for (int c = 0; c < C; ++c) acc = fmaf(fmaf(x[v * C + c], ssc[c], ssh[c]), sw[c * N + n], acc);
      pr[n] = acc;
      mx = fmaxf(mx, acc);
    }
    float den = 0.f;
    for (int n = 0; n < N; ++n) {
      const float e = expf(pr[n] - mx);
      pr[n] = e;
      den += e;
    }
    const float inv = 1.f / den;
    for (int n = 0; n < N; ++n) pr[n] *= inv;
  }
}

// Soft-Dice sums over the K generation labels that have an equivalent among the segmentation labels:
//   pred_k[v] = sum_j probs[v][cls_idx[k][j]] (up to 3 merged labels, -1 = unused),  gt_k[v] = (seg[v] == cls_gt[k])
//   sums[k] += 2 gt pred,  sums[K + k] += gt^2 + pred^2                                   (DiceLoss, layers.py:1343-1362)
__global__ __launch_bounds__(256) void seg_dice_sums_kernel(const float* __restrict__ probs, const int32_t* __restrict__ seg,
                                                            int64_t nvox, int N, const int32_t* __restrict__ cls_idx,
                                                            const int32_t* __restrict__ cls_gt, int K,
                                                            float* __restrict__ sums) {
  // per-wave partials in fixed LDS slots, added in wave order, then ONE flush per workgroup (syn_det_gather: in deterministic
  // mode the last workgroup adds all rows in id order) -- round 4: the LDS float atomics this kernel used made the Dice sums the
  // one reduction deterministic mode did not cover
  __shared__ float st[2 * SEG_MAXK], sw4[4][2 * SEG_MAXK];
  for (int k = 0; k < K; ++k) {
    const int i0 = cls_idx[3 * k], i1 = cls_idx[3 * k + 1], i2 = cls_idx[3 * k + 2], g = cls_gt[k];
    float top = 0.f, bot = 0.f;
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * blockDim.x) {
      const float* pr = probs + v * N;
      float p = pr[i0];
      if (i1 >= 0) p += pr[i1];
      if (i2 >= 0) p += pr[i2];
      const float gt = seg[v] == g ? 1.f : 0.f;
      top += 2.f * gt * p;
      bot += gt + p * p;
    }
    top = syn_wave_sum(top);
    bot = syn_wave_sum(bot);
    if ((threadIdx.x & 63) == 0) {
      sw4[threadIdx.x >> 6][k] = top;
      sw4[threadIdx.x >> 6][K + k] = bot;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * K; i += blockDim.x) st[i] = (sw4[0][i] + sw4[1][i]) + (sw4[2][i] + sw4[3][i]);
  __syncthreads();
  if (syn_det_gather(st, 2 * K))
    for (int i = threadIdx.x; i < 2 * K; i += blockDim.x) atomicAdd(&sums[i], st[i]);
  syn_det_gather_end(2 * K);
}

// Backward of  scale * mean_k(1 - (T_k + e)/(B_k + e))  through the label merging, the softmax and the 1x1x1 head:
//   dpred_k = -scale/K * (2 gt_k (B_k+e) - 2 pred_k (T_k+e)) / (B_k+e)^2
//   dlogit_n = p_n (dprob_n - S),  dprob_n = dpred_{class(n)} (0 for labels without class),  S = sum_k dpred_k pred_k
//   dbn[v][c] = sum_n W[c][n] dlogit_n  = sum_k dpred_k sum_j W[c][idx_kj] p_idx_kj  -  S sum_n W[c][n] p_n
__global__ __launch_bounds__(256) void seg_dice_bwd_kernel(const float* __restrict__ probs, const int32_t* __restrict__ seg,
                                                           int64_t nvox, int C, int N, const float* __restrict__ W,
                                                           const int32_t* __restrict__ cls_idx,
                                                           const int32_t* __restrict__ cls_gt, int K,
                                                           const float* __restrict__ sums, float scale,
                                                           float* __restrict__ dbn) {
  __shared__ float sw[SEG_MAXC * SEG_MAXN], sT[SEG_MAXK], sB[SEG_MAXK];
  __shared__ int sidx[3 * SEG_MAXK], sgt[SEG_MAXK];
  for (int i = threadIdx.x; i < C * N; i += blockDim.x) sw[i] = W[i];
  for (int i = threadIdx.x; i < K; i += blockDim.x) {
    sT[i] = sums[i] + 1e-7f;
    sB[i] = sums[K + i] + 1e-7f;
    sgt[i] = cls_gt[i];
  }
  for (int i = threadIdx.x; i < 3 * K; i += blockDim.x) sidx[i] = cls_idx[i];
  __syncthreads();
  const float coef = -scale / (float)K;
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * blockDim.x) {
    const float* pr = probs + v * N;
    float out[SEG_MAXC];
#pragma unroll
    for (int c = 0; c < SEG_MAXC; ++c) out[c] = 0.f;
    float S = 0.f;
    const int lab = seg[v];
    for (int k = 0; k < K; ++k) {
      const int i0 = sidx[3 * k], i1 = sidx[3 * k + 1], i2 = sidx[3 * k + 2];
      const float p0 = pr[i0], p1 = i1 >= 0 ? pr[i1] : 0.f, p2 = i2 >= 0 ? pr[i2] : 0.f;
      const float pk = p0 + p1 + p2;
      const float gt = lab == sgt[k] ? 1.f : 0.f;
      const float dp = coef * (2.f * gt * sB[k] - 2.f * pk * sT[k]) / (sB[k] * sB[k]);
      S += dp * pk;
#pragma unroll
      for (int c = 0; c < SEG_MAXC; ++c)
        if (c < C) {
          float a = sw[c * N + i0] * p0;
          if (i1 >= 0) a += sw[c * N + i1] * p1;
          if (i2 >= 0) a += sw[c * N + i2] * p2;
          out[c] += dp * a;
        }
    }
    for (int n = 0; n < N; ++n) {
      const float sp = S * pr[n];
#pragma unroll
      for (int c = 0; c < SEG_MAXC; ++c)
        if (c < C) out[c] -= sw[c * N + n] * sp;
    }
#pragma unroll
    for (int c = 0; c < SEG_MAXC; ++c)
      if (c < C) dbn[v * C + c] = out[c];
  }
}

// ------------------------------------------------------------------------------------------ Adam (Keras 2.3.1)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr_t,
                                                   float b1, float b2, float eps, float gs) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * gs;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
  }
}

inline bool ok_c4(int C) { return C > 0 && (C % 4) == 0 && C <= 4096; }
inline bool bad_shape(const int s[3]) { return s[0] <= 0 || s[1] <= 0 || s[2] <= 0; }
inline bool odd_shape(const int s[3]) { return (s[0] | s[1] | s[2]) & 1; }

}  // namespace

// ---- entry-point bodies, templated on the activation type (float | bf16_t)
template <typename T>
int elu_bwd_t(const T* dy, const T* dy2, const T* y, T* dz, float* dbias, int64_t nvox, int C, synthsr_stream_t stream) {
  if (!dy || !y || !dz || nvox < 1 || !ok_c4(C)) return SYNTHSR_EINVAL;
  const int64_t n4 = nvox * (C / 4);
  hipLaunchKernelGGL(elu_bwd_kernel<T>, dim3(syn_grid(n4, RB, red_grid())), dim3(RB), C * sizeof(float), (hipStream_t)stream, dy,
                     dy2, y, dz, dbias, n4, C / 4, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                     0.f, 0.f, (const float*)nullptr, (const float*)nullptr);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

template <typename T>
int bn_elu_bwd_t(const T* dy, const T* dy2, const T* y, T* dz, float* dbias, int64_t nvox, int C, const float* stats, const float* gamma, float eps, const float* sums, synthsr_stream_t stream) {
  if (!dy || !y || !dz || !stats || !gamma || !sums || nvox < 1 || !ok_c4(C)) return SYNTHSR_EINVAL;
  const int64_t n4 = nvox * (C / 4);
  hipLaunchKernelGGL(elu_bwd_kernel<T>, dim3(syn_grid(n4, RB, red_grid())), dim3(RB), C * sizeof(float), (hipStream_t)stream, dy,
                     dy2, y, dz, dbias, n4, C / 4, stats, gamma, sums, eps, (float)(1.0 / (double)nvox),
                     (const float*)nullptr, (const float*)nullptr);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

template <typename T>
int bn_elu_bwd_head_t(const float* dpred, const float* whead, const T* y, T* dz, float* dbias, int64_t nvox, int C, const float* stats, const float* gamma, float eps, const float* sums, synthsr_stream_t stream) {
  if (!dpred || !whead || !y || !dz || !stats || !gamma || !sums || nvox < 1 || !ok_c4(C)) return SYNTHSR_EINVAL;
  const int64_t n4 = nvox * (C / 4);
  hipLaunchKernelGGL(elu_bwd_kernel<T>, dim3(syn_grid(n4, RB, red_grid())), dim3(RB), C * sizeof(float), (hipStream_t)stream,
                     (const T*)nullptr, (const T*)nullptr, y, dz, dbias, n4, C / 4, stats, gamma, sums, eps,
                     (float)(1.0 / (double)nvox), dpred, whead);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

template <typename T>
int bn_stats_t(const T* x, int64_t nvox, int C, float* stats, double* ws, synthsr_stream_t stream) {
  if (!x || !stats || !ws || nvox < 1 || !ok_c4(C)) return SYNTHSR_EINVAL;
  if (hipMemsetAsync(ws, 0, 2 * C * sizeof(double), (hipStream_t)stream) != hipSuccess) return SYNTHSR_ELAUNCH;
  const int64_t n4 = nvox * (C / 4);
  hipLaunchKernelGGL(bn_stats_kernel<T>, dim3(syn_grid(n4, RB, red_grid())), dim3(RB), 2 * C * sizeof(float),
                     (hipStream_t)stream, x, n4, C / 4, ws);
  SYN_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, ws, stats, C,
                     1.0 / (double)nvox);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

template <typename T>
int bn_maxpool_t(const T* x, T* y, const int shape[3], int C, const float* stats, const float* gamma, const float* beta, float eps, synthsr_stream_t stream) {
  if (!x || !y || !stats || !gamma || !beta || bad_shape(shape) || odd_shape(shape) || !ok_c4(C)) return SYNTHSR_EINVAL;
  Shape3 s{{shape[0], shape[1], shape[2]}};
  const int64_t n4 = (int64_t)(s.d[0] / 2) * (s.d[1] / 2) * (s.d[2] / 2) * (C / 4);
  hipLaunchKernelGGL(bn_maxpool_kernel<T>, dim3(syn_grid(n4, 256)), dim3(256), 0, (hipStream_t)stream, x, y, s, C, stats,
                     gamma, beta, eps);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

template <typename T>
int bn_maxpool_bwd_t(const T* dy, const T* x, T* dbn, const int shape[3], int C, const float* stats, const float* gamma, const float* beta, float eps, synthsr_stream_t stream) {
  if (!dy || !x || !dbn || !stats || !gamma || !beta || bad_shape(shape) || odd_shape(shape) || !ok_c4(C))
    return SYNTHSR_EINVAL;
  Shape3 s{{shape[0], shape[1], shape[2]}};
  const int64_t n4 = (int64_t)(s.d[0] / 2) * (s.d[1] / 2) * (s.d[2] / 2) * (C / 4);
  hipLaunchKernelGGL(bn_maxpool_bwd_kernel<T>, dim3(syn_grid(n4, 256)), dim3(256), 0, (hipStream_t)stream, dy, x, dbn, s,
                     C, stats, gamma, beta, eps);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

template <typename T>
int bn_maxpool_bwd_ex_t(const T* dy, const T* x, T* dbn, const int shape[3], int C, const float* stats, const float* gamma, const float* beta, float eps, float* sums, synthsr_stream_t stream) {
  if (!sums) return bn_maxpool_bwd_t<T>(dy, x, dbn, shape, C, stats, gamma, beta, eps, stream);
  if (!dy || !x || !stats || !gamma || !beta || bad_shape(shape) || odd_shape(shape) || !ok_c4(C))  // dbn may be NULL: sums only
    return SYNTHSR_EINVAL;
  Shape3 s{{shape[0], shape[1], shape[2]}};
  const int64_t n4 = (int64_t)(s.d[0] / 2) * (s.d[1] / 2) * (s.d[2] / 2) * (C / 4);
  hipLaunchKernelGGL(bn_maxpool_bwd_sums_kernel<T>, dim3(syn_grid(n4, RB, red_grid())), dim3(RB), 2 * C * sizeof(float),
                     (hipStream_t)stream, dy, x, dbn, s, C, stats, gamma, beta, eps, sums);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

template <typename T>
int bn_pool_elu_bwd_t(const T* dpool, const T* y, const T* dy2, T* dz, float* dbias, const int shape[3], int C,
                      const float* stats, const float* gamma, const float* beta, const float* sums, float eps,
                      synthsr_stream_t stream) {
  if (!dpool || !y || !dz || !stats || !gamma || !beta || !sums || bad_shape(shape) || odd_shape(shape) || !ok_c4(C))
    return SYNTHSR_EINVAL;
  Shape3 s{{shape[0], shape[1], shape[2]}};
  const int64_t n4 = (int64_t)(s.d[0] / 2) * (s.d[1] / 2) * (s.d[2] / 2) * (C / 4);
  const float inv_n = 1.0f / (float)((int64_t)s.d[0] * s.d[1] * s.d[2]);
  hipLaunchKernelGGL(bn_pool_elu_bwd_kernel<T>, dim3(syn_grid(n4, RB, red_grid())), dim3(RB), C * sizeof(float),
                     (hipStream_t)stream, dpool, y, dy2, dz, dbias, s, C, stats, gamma, beta, sums, eps, inv_n);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

template <typename T>
int bn_bwd_reduce_t(const T* dy, const T* x, int64_t nvox, int C, const float* stats, float eps, float* sums, synthsr_stream_t stream) {
  if (!dy || !x || !stats || !sums || nvox < 1 || !ok_c4(C)) return SYNTHSR_EINVAL;
  const int64_t n4 = nvox * (C / 4);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel<T>, dim3(syn_grid(n4, RB, red_grid())), dim3(RB), 2 * C * sizeof(float),
                     (hipStream_t)stream, dy, x, n4, C, stats, eps, sums);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

template <typename T>
int upsample_concat_t(const T* skip, const T* lo, T* out, const int shape[3], int Cs, int Cl, const float* stats, const float* gamma, const float* beta, float eps, synthsr_stream_t stream) {
  if (!skip || !lo || !out || !stats || !gamma || !beta || bad_shape(shape) || odd_shape(shape) || !ok_c4(Cs) ||
      !ok_c4(Cl))
    return SYNTHSR_EINVAL;
  Shape3 s{{shape[0], shape[1], shape[2]}};
  const int64_t n4 = (int64_t)s.d[0] * s.d[1] * s.d[2] * ((Cs + Cl) / 4);
  hipLaunchKernelGGL(upsample_concat_kernel<T>, dim3(syn_grid(n4, 256)), dim3(256), 0, (hipStream_t)stream, skip, lo, out,
                     s, Cs, Cl, stats, gamma, beta, eps);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

template <typename T>
int upsample_concat_bwd_t(const T* dcat, T* dskip, T* dlo_bn, const int shape[3], int Cs, int Cl, synthsr_stream_t stream) {
  if (!dcat || !dskip || !dlo_bn || bad_shape(shape) || odd_shape(shape) || !ok_c4(Cs) || !ok_c4(Cl))
    return SYNTHSR_EINVAL;
  Shape3 s{{shape[0], shape[1], shape[2]}};
  const int64_t nv = (int64_t)s.d[0] * s.d[1] * s.d[2];
  const int64_t n4 = nv * (Cs / 4) + (nv / 8) * (Cl / 4);
  hipLaunchKernelGGL(upsample_concat_bwd_kernel<T>, dim3(syn_grid(n4, 256)), dim3(256), 0, (hipStream_t)stream, dcat,
                     dskip, dlo_bn, s, Cs, Cl);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

template <typename T>
int head_loss_fwd_t(const T* x, const int* shape, int C, const float* stats, const float* gamma, const float* beta, float eps, const float* w, const float* b, int K, const float* residual, int res_stride, const int* res_offs, const float* target, float* pred, float* dpred, float* loss, int kind, const int* crop, synthsr_stream_t stream, float* ab = nullptr) {
  if (!x || !shape || !stats || !gamma || !beta || !w || !b || !target || !loss || !ok_c4(C)) return SYNTHSR_EINVAL;
  if (ab && (K != 1 || kind == 2 || C > 120)) return SYNTHSR_EINVAL;  // the fused backward sums exist for the 1-channel l1 / l2 head
  if (shape[0] < 1 || shape[1] < 1 || shape[2] < 1) return SYNTHSR_EINVAL;
  if (kind < 0 || kind > 2 || K < 1 || K > 4 || (kind == 2 && (K & 1))) return SYNTHSR_EINVAL;
  const int NT = kind == 2 ? K / 2 : K;
  if (C > 116) return SYNTHSR_EINVAL;  // LDS: (2 + K) C + 256 (C + 4) floats
  const int64_t nvox = (int64_t)shape[0] * shape[1] * shape[2];
  HeadBox box;
  for (int k = 0; k < 4; ++k) box.res_off[k] = 0;
  if (residual) {
    if (res_stride < 1 || !res_offs) return SYNTHSR_EINVAL;
    for (int k = 0; k < NT; ++k) {
      if (res_offs[k] < 0 || res_offs[k] >= res_stride) return SYNTHSR_EINVAL;
      box.res_off[k] = res_offs[k];
    }
  }
  box.on = crop != nullptr;
  box.d1 = shape[1];
  box.d2 = shape[2];
  int64_t n_in = nvox;
  if (crop) {
    n_in = 1;
    for (int i = 0; i < 3; ++i) {
      if (crop[i] < 0 || crop[3 + i] < 1 || crop[i] + crop[3 + i] > shape[i]) return SYNTHSR_EINVAL;
      box.lo[i] = crop[i];
      box.hi[i] = crop[i] + crop[3 + i];
      n_in *= crop[3 + i];
    }
  } else {
    for (int i = 0; i < 3; ++i) {
      box.lo[i] = 0;
      box.hi[i] = shape[i];
    }
  }
  const size_t smem = ((2 + K) * C + 256 * (C + 4)) * sizeof(float);
  const float inv_n = (float)(1.0 / ((double)n_in * NT));
  const dim3 grid(syn_grid(nvox, 256, head_grid()));
#define SYN_HEAD_FWD(KK)                                                                                                 \
  hipLaunchKernelGGL((head_loss_fwd_kernel<T, KK>), grid, dim3(256), smem, (hipStream_t)stream, x, nvox, C, stats, gamma, beta, \
                     eps, w, b, residual, res_stride, target, pred, dpred, loss, inv_n, kind, box, ab)
  switch (K) {
    case 1:
      if (C == 24) {
        hipLaunchKernelGGL((head_loss_fwd_kernel<T, 1, 24>), grid, dim3(256), smem, (hipStream_t)stream, x, nvox, C, stats, gamma, beta,
                           eps, w, b, residual, res_stride, target, pred, dpred, loss, inv_n, kind, box, ab);
        break;
      }
      SYN_HEAD_FWD(1);
      break;
    case 2: SYN_HEAD_FWD(2); break;
    case 3: SYN_HEAD_FWD(3); break;
    default: SYN_HEAD_FWD(4); break;
  }
#undef SYN_HEAD_FWD
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

template <typename T>
int head_bwd_multi_t(const float* dpred, const T* x, int64_t nvox, int C, int K, const float* stats, const float* gamma, const float* beta, float eps, const float* w, T* dbn, float* dw, float* db, synthsr_stream_t stream) {
  if (!dpred || !x || !stats || !gamma || !beta || !w || !dbn || !dw || !db || nvox < 1 || !ok_c4(C) || K < 2 || K > 4)
    return SYNTHSR_EINVAL;
  const int64_t n4 = nvox * (C / 4);
  const dim3 grid(syn_grid(n4, RB, head_grid()));
  const size_t smem = (K * C + K) * sizeof(float);
#define SYN_HEAD_BWD(KK)                                                                                             \
  hipLaunchKernelGGL((head_multi_bwd_kernel<T, KK>), grid, dim3(RB), smem, (hipStream_t)stream, dpred, x, n4, C, stats, gamma, \
                     beta, eps, w, dbn, dw, db)
  switch (K) {
    case 2: SYN_HEAD_BWD(2); break;
    case 3: SYN_HEAD_BWD(3); break;
    default: SYN_HEAD_BWD(4); break;
  }
#undef SYN_HEAD_BWD
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

template <typename T>
int head_bwd_ex_t(const float* dpred, const T* x, int64_t nvox, int C, const float* stats, const float* gamma, const float* beta, float eps, const float* w, T* dbn, float* dw, float* db, float* bn_sums, synthsr_stream_t stream) {
  if (!dpred || !x || !stats || !gamma || !beta || !w || !dw || !db || nvox < 1 || !ok_c4(C)) return SYNTHSR_EINVAL;
  const int64_t n4 = nvox * (C / 4);
  hipLaunchKernelGGL(head_bwd_kernel<T>, dim3(syn_grid(n4, RB, red_grid())), dim3(RB), (C + 1) * sizeof(float),
                     (hipStream_t)stream, dpred, x, n4, C, stats, gamma, beta, eps, w, dbn, dw, db, bn_sums);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

template <typename T>
int head_bwd_t(const float* dpred, const T* x, int64_t nvox, int C, const float* stats, const float* gamma, const float* beta, float eps, const float* w, T* dbn, float* dw, float* db, synthsr_stream_t stream) {
  if (!dpred || !x || !stats || !gamma || !beta || !w || !dbn || !dw || !db || nvox < 1 || !ok_c4(C))
    return SYNTHSR_EINVAL;
  const int64_t n4 = nvox * (C / 4);
  hipLaunchKernelGGL(head_bwd_kernel<T>, dim3(syn_grid(n4, RB, red_grid())), dim3(RB), (C + 1) * sizeof(float),
                     (hipStream_t)stream, dpred, x, n4, C, stats, gamma, beta, eps, w, dbn, dw, db, (float*)nullptr);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

template <typename T>
static int elu_bwd_drop_t(const T* dy, const T* dy2, const T* y, T* dz, float* dbias, int64_t nvox, int C, const float* stats,
                          const float* gamma, float eps, const float* sums, const float* dpred, const float* whead,
                          const float* drop, int64_t nvox_per_sample, synthsr_stream_t stream) {
  const bool bn = stats || gamma || sums, head = dpred || whead;
  if (!y || !dz || !drop || nvox < 1 || !ok_c4(C) || nvox_per_sample < 1 || (nvox % nvox_per_sample) != 0 ||
      (bn && (!stats || !gamma || !sums)) || (head && (!dpred || !whead || !bn || dy || dy2)) || (!head && !dy))
    return SYNTHSR_EINVAL;
  const int64_t n4 = nvox * (C / 4);
  hipLaunchKernelGGL(elu_bwd_kernel<T>, dim3(syn_grid(n4, RB, red_grid())), dim3(RB), C * sizeof(float), (hipStream_t)stream,
                     dy, dy2, y, dz, dbias, n4, C / 4, stats, gamma, sums, eps, bn ? (float)(1.0 / (double)nvox) : 0.f, dpred,
                     whead, drop, nvox_per_sample * (C / 4));
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

extern "C" {

int synthsr_elu_bwd(const float* dy, const float* dy2, const float* y, float* dz, float* dbias, int64_t nvox, int C, synthsr_stream_t stream) {
  return elu_bwd_t<float>(dy, dy2, y, dz, dbias, nvox, C, stream);
}
int synthsr_elu_bwd_bf16(const void* dy, const void* dy2, const void* y, void* dz, float* dbias, int64_t nvox, int C, synthsr_stream_t stream) {
  return elu_bwd_t<bf16_t>((const bf16_t*)dy, (const bf16_t*)dy2, (const bf16_t*)y, (bf16_t*)dz, dbias, nvox, C, stream);
}

int synthsr_bn_elu_bwd(const float* dy, const float* dy2, const float* y, float* dz, float* dbias, int64_t nvox, int C, const float* stats, const float* gamma, float eps, const float* sums, synthsr_stream_t stream) {
  return bn_elu_bwd_t<float>(dy, dy2, y, dz, dbias, nvox, C, stats, gamma, eps, sums, stream);
}
int synthsr_bn_elu_bwd_bf16(const void* dy, const void* dy2, const void* y, void* dz, float* dbias, int64_t nvox, int C, const float* stats, const float* gamma, float eps, const float* sums, synthsr_stream_t stream) {
  return bn_elu_bwd_t<bf16_t>((const bf16_t*)dy, (const bf16_t*)dy2, (const bf16_t*)y, (bf16_t*)dz, dbias, nvox, C, stats, gamma, eps, sums, stream);
}

int synthsr_bn_elu_bwd_head(const float* dpred, const float* whead, const float* y, float* dz, float* dbias, int64_t nvox, int C, const float* stats, const float* gamma, float eps, const float* sums, synthsr_stream_t stream) {
  return bn_elu_bwd_head_t<float>(dpred, whead, y, dz, dbias, nvox, C, stats, gamma, eps, sums, stream);
}
int synthsr_bn_elu_bwd_head_bf16(const float* dpred, const float* whead, const void* y, void* dz, float* dbias, int64_t nvox, int C, const float* stats, const float* gamma, float eps, const float* sums, synthsr_stream_t stream) {
  return bn_elu_bwd_head_t<bf16_t>(dpred, whead, (const bf16_t*)y, (bf16_t*)dz, dbias, nvox, C, stats, gamma, eps, sums, stream);
}

int synthsr_scale_channels(const float* x, float* out, int64_t nvox, int C, const float* scale, int64_t nvox_per_sample,
                           synthsr_stream_t stream) {
  if (!x || !out || !scale || nvox < 1 || !ok_c4(C) || nvox_per_sample < 1 || (nvox % nvox_per_sample) != 0) return SYNTHSR_EINVAL;
  const int64_t n4 = nvox * (C / 4);
  hipLaunchKernelGGL(scale_channels_kernel<float>, dim3(syn_grid(n4, 256)), dim3(256), 0, (hipStream_t)stream, x, out, n4, C / 4,
                     scale, nvox_per_sample * (C / 4));
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}
int synthsr_scale_channels_bf16(const void* x, void* out, int64_t nvox, int C, const float* scale, int64_t nvox_per_sample,
                                synthsr_stream_t stream) {
  if (!x || !out || !scale || nvox < 1 || !ok_c4(C) || nvox_per_sample < 1 || (nvox % nvox_per_sample) != 0) return SYNTHSR_EINVAL;
  const int64_t n4 = nvox * (C / 4);
  hipLaunchKernelGGL(scale_channels_kernel<bf16_t>, dim3(syn_grid(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)out, n4, C / 4, scale, nvox_per_sample * (C / 4));
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_elu_bwd_drop(const float* dy, const float* dy2, const float* y, float* dz, float* dbias, int64_t nvox, int C,
                         const float* stats, const float* gamma, float eps, const float* sums, const float* dpred,
                         const float* whead, const float* drop, int64_t nvox_per_sample, synthsr_stream_t stream) {
  return elu_bwd_drop_t<float>(dy, dy2, y, dz, dbias, nvox, C, stats, gamma, eps, sums, dpred, whead, drop, nvox_per_sample, stream);
}
int synthsr_elu_bwd_drop_bf16(const void* dy, const void* dy2, const void* y, void* dz, float* dbias, int64_t nvox, int C,
                              const float* stats, const float* gamma, float eps, const float* sums, const float* dpred,
                              const float* whead, const float* drop, int64_t nvox_per_sample, synthsr_stream_t stream) {
  return elu_bwd_drop_t<bf16_t>((const bf16_t*)dy, (const bf16_t*)dy2, (const bf16_t*)y, (bf16_t*)dz, dbias, nvox, C, stats, gamma,
                                eps, sums, dpred, whead, drop, nvox_per_sample, stream);
}

int synthsr_bn_stats(const float* x, int64_t nvox, int C, float* stats, double* ws, synthsr_stream_t stream) {
  return bn_stats_t<float>(x, nvox, C, stats, ws, stream);
}
int synthsr_bn_stats_bf16(const void* x, int64_t nvox, int C, float* stats, double* ws, synthsr_stream_t stream) {
  return bn_stats_t<bf16_t>((const bf16_t*)x, nvox, C, stats, ws, stream);
}

// mean / variance from per-workgroup partial sums: partial[nwg][2C] (sum | sum of squares), accumulated in double
__global__ void bn_stats_partials_kernel(const float* __restrict__ partial, int nwg, float* __restrict__ stats, int C,
                                         double inv_n) {
  const int c = blockIdx.x;  // one block per channel
  double a = 0.0, q = 0.0;
  for (int b = threadIdx.x; b < nwg; b += blockDim.x) {
    a += (double)partial[(size_t)b * 2 * C + c];
    q += (double)partial[(size_t)b * 2 * C + C + c];
  }
  __shared__ double sa[64], sq[64];
  sa[threadIdx.x] = a;
  sq[threadIdx.x] = q;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 64; ++i) {
      a += sa[i];
      q += sq[i];
    }
    const double m = a * inv_n;
    double v = q * inv_n - m * m;
    if (v < 0.0) v = 0.0;
    stats[c] = (float)m;
    stats[C + c] = (float)v;
  }
}

int synthsr_bn_stats_from_partials(const float* partial, int nwg, int64_t nvox, int C, float* stats,
                                   synthsr_stream_t stream) {
  if (!partial || !stats || nwg < 1 || nvox < 1 || C < 1) return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(bn_stats_partials_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, partial, nwg, stats, C,
                     1.0 / (double)nvox);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_bn_apply(const float* x, float* y, int64_t nvox, int C, const float* stats, const float* gamma,
                     const float* beta, float eps, synthsr_stream_t stream) {
  if (!x || !y || !stats || !gamma || !beta || nvox < 1 || !ok_c4(C)) return SYNTHSR_EINVAL;
  const int64_t n4 = nvox * (C / 4);
  hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(syn_grid(n4, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n4, C, stats,
                     gamma, beta, eps);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}
int synthsr_bn_apply_bf16(const void* x, void* y, int64_t nvox, int C, const float* stats, const float* gamma,
                          const float* beta, float eps, synthsr_stream_t stream) {
  if (!x || !y || !stats || !gamma || !beta || nvox < 1 || !ok_c4(C)) return SYNTHSR_EINVAL;
  const int64_t n4 = nvox * (C / 4);
  hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(syn_grid(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)y, n4, C, stats, gamma, beta, eps);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_bn_maxpool(const float* x, float* y, const int shape[3], int C, const float* stats, const float* gamma, const float* beta, float eps, synthsr_stream_t stream) {
  return bn_maxpool_t<float>(x, y, shape, C, stats, gamma, beta, eps, stream);
}
int synthsr_bn_maxpool_bf16(const void* x, void* y, const int shape[3], int C, const float* stats, const float* gamma, const float* beta, float eps, synthsr_stream_t stream) {
  return bn_maxpool_t<bf16_t>((const bf16_t*)x, (bf16_t*)y, shape, C, stats, gamma, beta, eps, stream);
}

int synthsr_bn_maxpool_bwd(const float* dy, const float* x, float* dbn, const int shape[3], int C, const float* stats, const float* gamma, const float* beta, float eps, synthsr_stream_t stream) {
  return bn_maxpool_bwd_t<float>(dy, x, dbn, shape, C, stats, gamma, beta, eps, stream);
}
int synthsr_bn_maxpool_bwd_bf16(const void* dy, const void* x, void* dbn, const int shape[3], int C, const float* stats, const float* gamma, const float* beta, float eps, synthsr_stream_t stream) {
  return bn_maxpool_bwd_t<bf16_t>((const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)dbn, shape, C, stats, gamma, beta, eps, stream);
}

int synthsr_bn_maxpool_bwd_ex(const float* dy, const float* x, float* dbn, const int shape[3], int C, const float* stats, const float* gamma, const float* beta, float eps, float* sums, synthsr_stream_t stream) {
  return bn_maxpool_bwd_ex_t<float>(dy, x, dbn, shape, C, stats, gamma, beta, eps, sums, stream);
}
int synthsr_bn_pool_elu_bwd(const float* dpool, const float* y, const float* dy2, float* dz, float* dbias, const int shape[3],
                            int C, const float* stats, const float* gamma, const float* beta, const float* sums, float eps,
                            synthsr_stream_t stream) {
  return bn_pool_elu_bwd_t<float>(dpool, y, dy2, dz, dbias, shape, C, stats, gamma, beta, sums, eps, stream);
}
int synthsr_bn_pool_elu_bwd_bf16(const void* dpool, const void* y, const void* dy2, void* dz, float* dbias, const int shape[3],
                                 int C, const float* stats, const float* gamma, const float* beta, const float* sums, float eps,
                                 synthsr_stream_t stream) {
  return bn_pool_elu_bwd_t<bf16_t>((const bf16_t*)dpool, (const bf16_t*)y, (const bf16_t*)dy2, (bf16_t*)dz, dbias, shape, C, stats,
                                   gamma, beta, sums, eps, stream);
}
int synthsr_bn_maxpool_bwd_ex_bf16(const void* dy, const void* x, void* dbn, const int shape[3], int C, const float* stats, const float* gamma, const float* beta, float eps, float* sums, synthsr_stream_t stream) {
  return bn_maxpool_bwd_ex_t<bf16_t>((const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)dbn, shape, C, stats, gamma, beta, eps, sums, stream);
}

int synthsr_bn_bwd_reduce(const float* dy, const float* x, int64_t nvox, int C, const float* stats, float eps, float* sums, synthsr_stream_t stream) {
  return bn_bwd_reduce_t<float>(dy, x, nvox, C, stats, eps, sums, stream);
}
int synthsr_bn_bwd_reduce_bf16(const void* dy, const void* x, int64_t nvox, int C, const float* stats, float eps, float* sums, synthsr_stream_t stream) {
  return bn_bwd_reduce_t<bf16_t>((const bf16_t*)dy, (const bf16_t*)x, nvox, C, stats, eps, sums, stream);
}

int synthsr_bn_bwd_apply(const float* dy, const float* x, float* dx, int64_t nvox, int C, const float* stats,
                         const float* gamma, float eps, const float* sums, synthsr_stream_t stream) {
  if (!dy || !x || !dx || !stats || !gamma || !sums || nvox < 1 || !ok_c4(C)) return SYNTHSR_EINVAL;
  const int64_t n4 = nvox * (C / 4);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(syn_grid(n4, 256)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, n4, C,
                     stats, gamma, eps, sums, (float)(1.0 / (double)nvox));
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_upsample_concat(const float* skip, const float* lo, float* out, const int shape[3], int Cs, int Cl, const float* stats, const float* gamma, const float* beta, float eps, synthsr_stream_t stream) {
  return upsample_concat_t<float>(skip, lo, out, shape, Cs, Cl, stats, gamma, beta, eps, stream);
}
int synthsr_upsample_concat_bf16(const void* skip, const void* lo, void* out, const int shape[3], int Cs, int Cl, const float* stats, const float* gamma, const float* beta, float eps, synthsr_stream_t stream) {
  return upsample_concat_t<bf16_t>((const bf16_t*)skip, (const bf16_t*)lo, (bf16_t*)out, shape, Cs, Cl, stats, gamma, beta, eps, stream);
}

int synthsr_upsample_concat_bwd(const float* dcat, float* dskip, float* dlo_bn, const int shape[3], int Cs, int Cl, synthsr_stream_t stream) {
  return upsample_concat_bwd_t<float>(dcat, dskip, dlo_bn, shape, Cs, Cl, stream);
}
int synthsr_upsample_concat_bwd_bf16(const void* dcat, void* dskip, void* dlo_bn, const int shape[3], int Cs, int Cl, synthsr_stream_t stream) {
  return upsample_concat_bwd_t<bf16_t>((const bf16_t*)dcat, (bf16_t*)dskip, (bf16_t*)dlo_bn, shape, Cs, Cl, stream);
}

int synthsr_head_loss_fwd(const float* x, const int* shape, int C, const float* stats, const float* gamma, const float* beta, float eps, const float* w, const float* b, int K, const float* residual, int res_stride, const int* res_offs, const float* target, float* pred, float* dpred, float* loss, int kind, const int* crop, synthsr_stream_t stream) {
  return head_loss_fwd_t<float>(x, shape, C, stats, gamma, beta, eps, w, b, K, residual, res_stride, res_offs, target, pred, dpred, loss, kind, crop, stream);
}
int synthsr_head_loss_fwd_ab(const float* x, const int* shape, int C, const float* stats, const float* gamma, const float* beta, float eps, const float* w, const float* b, const float* residual, int res_stride, int res_off, const float* target, float* pred, float* dpred, float* loss, int kind, const int* crop, float* ab, synthsr_stream_t stream) {
  if (!ab) return SYNTHSR_EINVAL;
  return head_loss_fwd_t<float>(x, shape, C, stats, gamma, beta, eps, w, b, 1, residual, res_stride, &res_off, target, pred, dpred, loss, kind, crop, stream, ab);
}
int synthsr_head_loss_fwd_ab_bf16(const void* x, const int* shape, int C, const float* stats, const float* gamma, const float* beta, float eps, const float* w, const float* b, const float* residual, int res_stride, int res_off, const float* target, float* pred, float* dpred, float* loss, int kind, const int* crop, float* ab, synthsr_stream_t stream) {
  if (!ab) return SYNTHSR_EINVAL;
  return head_loss_fwd_t<bf16_t>((const bf16_t*)x, shape, C, stats, gamma, beta, eps, w, b, 1, residual, res_stride, &res_off, target, pred, dpred, loss, kind, crop, stream, ab);
}
int synthsr_head_bwd_from_sums(const float* ab, int C, const float* gamma, const float* beta, const float* w, float* dw, float* db, float* bn_sums, synthsr_stream_t stream) {
  if (!ab || !gamma || !beta || !w || !dw || !db || C < 1) return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(head_ab_finish_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, ab, C, gamma, beta, w, dw, db, bn_sums);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}
int synthsr_head_loss_fwd_bf16(const void* x, const int* shape, int C, const float* stats, const float* gamma, const float* beta, float eps, const float* w, const float* b, int K, const float* residual, int res_stride, const int* res_offs, const float* target, float* pred, float* dpred, float* loss, int kind, const int* crop, synthsr_stream_t stream) {
  return head_loss_fwd_t<bf16_t>((const bf16_t*)x, shape, C, stats, gamma, beta, eps, w, b, K, residual, res_stride, res_offs, target, pred, dpred, loss, kind, crop, stream);
}

int synthsr_head_l1_fwd(const float* x, int64_t nvox, int C, const float* stats, const float* gamma, const float* beta,
                        float eps, const float* w, const float* b, const float* residual, int res_stride, int res_off,
                        const float* target, float* pred, float* dpred, float* loss, synthsr_stream_t stream) {
  if (nvox < 1 || nvox >= (1ll << 31)) return SYNTHSR_EINVAL;
  const int shape[3] = {1, 1, (int)nvox};
  return synthsr_head_loss_fwd(x, shape, C, stats, gamma, beta, eps, w, b, 1, residual, res_stride, &res_off, target, pred,
                               dpred, loss, 0, nullptr, stream);
}

int synthsr_head_bwd_multi(const float* dpred, const float* x, int64_t nvox, int C, int K, const float* stats, const float* gamma, const float* beta, float eps, const float* w, float* dbn, float* dw, float* db, synthsr_stream_t stream) {
  return head_bwd_multi_t<float>(dpred, x, nvox, C, K, stats, gamma, beta, eps, w, dbn, dw, db, stream);
}
int synthsr_head_bwd_multi_bf16(const float* dpred, const void* x, int64_t nvox, int C, int K, const float* stats, const float* gamma, const float* beta, float eps, const float* w, void* dbn, float* dw, float* db, synthsr_stream_t stream) {
  return head_bwd_multi_t<bf16_t>(dpred, (const bf16_t*)x, nvox, C, K, stats, gamma, beta, eps, w, (bf16_t*)dbn, dw, db, stream);
}

int synthsr_head_bwd_ex(const float* dpred, const float* x, int64_t nvox, int C, const float* stats, const float* gamma, const float* beta, float eps, const float* w, float* dbn, float* dw, float* db, float* bn_sums, synthsr_stream_t stream) {
  return head_bwd_ex_t<float>(dpred, x, nvox, C, stats, gamma, beta, eps, w, dbn, dw, db, bn_sums, stream);
}
int synthsr_head_bwd_ex_bf16(const float* dpred, const void* x, int64_t nvox, int C, const float* stats, const float* gamma, const float* beta, float eps, const float* w, void* dbn, float* dw, float* db, float* bn_sums, synthsr_stream_t stream) {
  return head_bwd_ex_t<bf16_t>(dpred, (const bf16_t*)x, nvox, C, stats, gamma, beta, eps, w, (bf16_t*)dbn, dw, db, bn_sums, stream);
}

int synthsr_head_bwd(const float* dpred, const float* x, int64_t nvox, int C, const float* stats, const float* gamma, const float* beta, float eps, const float* w, float* dbn, float* dw, float* db, synthsr_stream_t stream) {
  return head_bwd_t<float>(dpred, x, nvox, C, stats, gamma, beta, eps, w, dbn, dw, db, stream);
}
int synthsr_head_bwd_bf16(const float* dpred, const void* x, int64_t nvox, int C, const float* stats, const float* gamma, const float* beta, float eps, const float* w, void* dbn, float* dw, float* db, synthsr_stream_t stream) {
  return head_bwd_t<bf16_t>(dpred, (const bf16_t*)x, nvox, C, stats, gamma, beta, eps, w, (bf16_t*)dbn, dw, db, stream);
}

int synthsr_seg_head_fwd(const float* x, int64_t nvox, int C, const float* stats, const float* gamma, const float* beta,
                         float eps, const float* w, const float* b, int N, float* probs, synthsr_stream_t stream) {
  if (!x || !stats || !gamma || !beta || !w || !b || !probs || nvox < 1 || C < 1 || C > SEG_MAXC || N < 1 || N > SEG_MAXN)
    return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(seg_head_fwd_kernel, dim3(syn_grid(nvox, 256)), dim3(256), 0, (hipStream_t)stream, x, nvox, C, stats,
                     gamma, beta, eps, w, b, N, probs);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_seg_dice_sums(const float* probs, const int32_t* seg, int64_t nvox, int N, const int32_t* cls_idx,
                          const int32_t* cls_gt, int K, float* sums, synthsr_stream_t stream) {
  if (!probs || !seg || !cls_idx || !cls_gt || !sums || nvox < 1 || N < 1 || N > SEG_MAXN || K < 1 || K > SEG_MAXK)
    return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(seg_dice_sums_kernel, dim3(syn_grid(nvox, 256, 1024)), dim3(256), 0, (hipStream_t)stream, probs, seg,
                     nvox, N, cls_idx, cls_gt, K, sums);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_seg_dice_bwd(const float* probs, const int32_t* seg, int64_t nvox, int C, int N, const float* w,
                         const int32_t* cls_idx, const int32_t* cls_gt, int K, const float* sums, float scale, float* dbn,
                         synthsr_stream_t stream) {
  if (!probs || !seg || !w || !cls_idx || !cls_gt || !sums || !dbn || nvox < 1 || C < 1 || C > SEG_MAXC || N < 1 ||
      N > SEG_MAXN || K < 1 || K > SEG_MAXK)
    return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(seg_dice_bwd_kernel, dim3(syn_grid(nvox, 256)), dim3(256), 0, (hipStream_t)stream, probs, seg, nvox, C,
                     N, w, cls_idx, cls_gt, K, sums, scale, dbn);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                      float eps, float grad_scale, synthsr_stream_t stream) {
  if (!p || !g || !m || !v || n < 1) return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(adam_kernel, dim3(syn_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_t, beta1,
                     beta2, eps, grad_scale);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

}  // extern "C"
