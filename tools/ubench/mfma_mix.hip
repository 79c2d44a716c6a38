// micro-benchmark: what the matrix pipe sustains (v_mfma_f32_16x16x32_bf16, 2 waves per SIMD) next to the other instruction
// classes of one K step of the split-arithmetic conv kernels (csrc/conv_split.hip: 48 MFMAs = 6 products x 4 rows x 2 co-tiles):
//   bit 0: 12 ds_read_b128 (activation fragments)     bit 1: 6 global_load_b128 from a 129 KB L2-resident set (weight fragments)
//   bit 2: 24 dependent-chain VALU (conversion / ELU)  bit 3: 3 ds_write_b64      bit 4: 1 global store b128 per step
//   hipcc --offload-arch=gfx950 -O3 -o mfma_mix mfma_mix.hip && ./mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, const u32x4* __restrict__ wts, int iters) {
  __shared__ u32x4 lds[2048];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2048; i += 256) lds[i] = wts[i];
  __syncthreads();
  f32x4 acc[4][2];
#pragma unroll
  for (int y = 0; y < 4; ++y)
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[y][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 wa[3][2], xb[3][4];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
#pragma unroll
    for (int m = 0; m < 2; ++m) wa[q][m] = lds[(tid + 64 * (q * 2 + m)) & 2047];
#pragma unroll
    for (int y = 0; y < 4; ++y) xb[q][y] = lds[(tid * 3 + 128 * (q * 4 + y)) & 2047];
  }
  float v0 = (float)tid, v1 = 1.f, v2 = 2.f;
  u32x4 st = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    u32x4 wn[3][2], xn[3][4];
    if (MODE & 1) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int y = 0; y < 4; ++y) xn[q][y] = lds[(tid * 3 + 128 * (q * 4 + y) + it) & 2047];
    }
    if (MODE & 2) {
      const u32x4* w = wts + ((it * 6) & 127) * 64 + lane;  // 8192 x 16 B = 128 KB set, walked fragment by fragment
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int m = 0; m < 2; ++m) wn[q][m] = w[(q * 2 + m) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int qa = c < 3 ? 0 : (c < 5 ? 1 : 2), qb = c == 0 ? 2 : (c == 1 ? 1 : (c == 2 ? 0 : (c == 3 ? 0 : (c == 4 ? 1 : 0))));
#pragma unroll
      for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int m = 0; m < 2; ++m)
          acc[y][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[qa][m]), __builtin_bit_cast(bf16x8, xb[qb][y]),
                                                              acc[y][m], 0, 0, 0);
      if (MODE & 4) {  // 4 dependent VALU per product group
        v0 = fmaf(v0, v1, v2);
        v1 = fmaf(v1, v2, v0);
        v2 = fmaf(v2, v0, v1);
        v0 = fmaf(v0, v2, v1);
      }
      if ((MODE & 8) && c < 3) *reinterpret_cast<uint2*>(&lds[(tid + c * 256 + it) & 2047]) = make_uint2(__float_as_uint(v0), it);
      if ((MODE & 16) && c == 5) {
        st[0] = __float_as_uint(v0) + it;
        *reinterpret_cast<u32x4*>(out + ((size_t)(blockIdx.x * 256 + tid) * 4 + ((size_t)(it & 63) << 22))) = st;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE & 1) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int y = 0; y < 4; ++y) xb[q][y] = xn[q][y];
    }
    if (MODE & 2) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int m = 0; m < 2; ++m) wa[q][m] = wn[q][m];
    }
  }
  float s = v0 + v1 + v2;
#pragma unroll
  for (int y = 0; y < 4; ++y)
#pragma unroll
    for (int m = 0; m < 2; ++m) s += acc[y][m][0] + acc[y][m][1] + acc[y][m][2] + acc[y][m][3];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, float* out, u32x4* in, int iters) {
  const int wgs = 512;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(wgs), dim3(256), 0, 0, out, in, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(wgs), dim3(256), 0, 0, out, in, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)wgs * 4 * iters * 48 * (2.0 * 16 * 16 * 32);
  printf("mode %2d  %-64s %.3f ms  %5.0f TFLOP/s  %.1f cycles/MFMA/SIMD at 2.4 GHz\n", MODE, name, ms, fl / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (2.0 * iters * 48));
}

int main() {
  float* out;
  u32x4* in;
  hipMalloc(&out, ((size_t)64 << 22) * 4 + 512 * 256 * 16);
  hipMalloc(&in, 8192 * sizeof(u32x4));
  hipMemset(in, 0, 8192 * sizeof(u32x4));
  const int iters = 4000;
  run<0>("48 MFMAs per step only", out, in, iters);
  run<1>("+ 12 ds_read_b128", out, in, iters);
  run<2>("+ 6 global_load_b128 (L2)", out, in, iters);
  run<3>("+ 12 ds_read_b128 + 6 global_load_b128", out, in, iters);
  run<4>("+ 24 VALU", out, in, iters);
  run<7>("+ ds_read + global_load + 24 VALU", out, in, iters);
  run<15>("+ ds_read + global_load + 24 VALU + 3 ds_write_b64", out, in, iters);
  run<31>("+ ds_read + global_load + 24 VALU + 3 ds_write_b64 + 1 store b128", out, in, iters);
  run<17>("+ 12 ds_read_b128 + 1 store b128", out, in, iters);
  return 0;
}
