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
