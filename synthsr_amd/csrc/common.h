// shared helpers for the gfx950 kernels of libsynthsr_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/synthsr_hip.h"

#define SYN_CHECK_LAUNCH()                                 \
  do {                                                     \
    hipError_t e__ = hipGetLastError();                    \
    if (e__ != hipSuccess) return SYNTHSR_ELAUNCH;         \
  } while (0)

static inline int64_t syn_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// grid for a grid-stride elementwise kernel: enough workgroups to fill 256 CUs several times over
static inline int syn_grid(int64_t n, int block, int max_blocks = 256 * 16) {
  int64_t g = syn_cdiv(n, block);
  if (g > max_blocks) g = max_blocks;
  if (g < 1) g = 1;
  return (int)g;
}

#ifdef __HIPCC__
// Contiguous element range [lo, hi) of this workgroup for a 1-D sweep over n elements, laid out XCD-contiguously:
// consecutive workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2), so XCD k gets ids k, k+8, ...;
// mapping id -> position (id & 7) * G/8 + id/8 gives every XCD ONE contiguous eighth of the volume instead of a comb
// over all of it (a gather / stencil sweep otherwise pulls the whole input through all 8 L2s: measured 7.3x the label
// volume in FETCH_SIZE for deform_gmm_kernel).  Chunks are multiples of 256 elements (block size): coalescing unchanged.
__device__ static inline void syn_block_range(int64_t n, int64_t& lo, int64_t& hi) {
  const int G = (int)gridDim.x, b = (int)blockIdx.x;
  const int pos = (G % 8 == 0) ? (b & 7) * (G >> 3) + (b >> 3) : b;
  const int64_t chunk = (((n + G - 1) / G) + 255) / 256 * 256;
  lo = (int64_t)pos * chunk;
  hi = lo + chunk < n ? lo + chunk : n;
}
#endif

// monotone uint32 encoding of float (total order incl. negatives) for atomicMin/atomicMax
__host__ __device__ static inline uint32_t syn_f2ord(float f) {
  union { float f; uint32_t u; } c;
  c.f = f;
  return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}
__host__ __device__ static inline float syn_ord2f(uint32_t u) {
  union { float f; uint32_t u; } c;
  c.u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return c.f;
}

// 64-lane wave reductions (CDNA wavefront = 64)
__device__ static inline float syn_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ static inline float syn_wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ static inline float syn_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
