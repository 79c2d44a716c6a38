// micro-benchmark: one K step (4 taps x 8 input channels) of the STACKED 24-output-channel split conv for the
// 64 voxels of a wave, six partial products, on the two bf16 MFMA shapes of gfx950 -- with the real instruction mix around it:
//   A  v_mfma_f32_16x16x32_bf16 (what csrc/conv_split.hip ships): weight rows stacked in 5 tiles of 16 (T0..T4), products b0 x {T0..T4},
//      b1 x {T0, T1, T3}, b2 x {T0, T3} = 10 MFMAs per 16-voxel row x 4 rows = 40 MFMAs of 16 cycles; operands: 12 ds_read_b128
//      (3 pieces x 4 rows), 5 weight fragments (L2)
//   B  v_mfma_f32_32x32x16_bf16: 32 voxels x 32 stacked rows x K 16.  The 72 weight rows (3 pieces x 24 channels) fill 3 tiles of
//      32 (2.25 used); b0 needs all three, b1 two, b2 one = 6 tile products x 2 column blocks of 32 voxels x 2 K halves = 24 MFMAs
//      of 32 cycles (768 vs 640 matrix cycles: 32-row tiles hold 144 useful of 192 stacked rows, 16-row tiles 144 of 160);
//      operands: 3 pieces x 2 column blocks x 2 K halves = 12 ds_read_b128 (the SAME LDS traffic: a 1 KB B fragment feeds 6 / 4 / 2
//      tile-cycles either way), 3 tiles x 2 K halves = 6 weight fragments, 96 instead of 80 accumulator registers
// Both with: 24 dependent VALU (the next chunk's conversion), 3 ds_write_b64, 1 store b128 per step.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip && ./mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int MIX>
__global__ __launch_bounds__(256, 2) void k(float* out, const u32x4* __restrict__ wts, int iters) {
  __shared__ u32x4 lds[2048];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2048; i += 256) lds[i] = wts[i];
  __syncthreads();
  constexpr int NW = SHAPE == 0 ? 5 : 6;  // weight fragments per step
  f32x4 accA[4][5];
  f32x16 accB[2][3];
#pragma unroll
  for (int y = 0; y < 4; ++y)
#pragma unroll
    for (int t = 0; t < 5; ++t) accA[y][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) accB[c][t][i] = 0.f;
  u32x4 wa[NW], xb[12];
#pragma unroll
  for (int q = 0; q < NW; ++q) wa[q] = lds[(tid + 64 * q) & 2047];
#pragma unroll
  for (int q = 0; q < 12; ++q) xb[q] = lds[(tid * 3 + 128 * q) & 2047];
  float v0 = (float)tid, v1 = 1.f, v2 = 2.f;
  u32x4 st = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    u32x4 wn[NW], xn[12];
    if (MIX) {
#pragma unroll
      for (int q = 0; q < 12; ++q) xn[q] = lds[(tid * 3 + 128 * q + it) & 2047];
      const u32x4* w = wts + ((it * NW) & 127) * 64 + lane;
#pragma unroll
      for (int q = 0; q < NW; ++q) wn[q] = w[q * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SHAPE == 0) {
      // xb[3 y + p] = piece p of row y; tiles: p 0 -> T0..T4, p 1 -> T0 T1 T3, p 2 -> T0 T3 (smallest first)
#pragma unroll
      for (int g = 0; g < 6; ++g) {
        constexpr int TL[6][2] = {{0, 2}, {3, 2}, {0, 1}, {3, 1}, {0, 0}, {3, 0}};
#pragma unroll
        for (int y = 0; y < 4; ++y)
          accA[y][TL[g][0]] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[TL[g][0]]),
                                                                      __builtin_bit_cast(bf16x8, xb[3 * y + TL[g][1]]), accA[y][TL[g][0]], 0, 0, 0);
        if (MIX) {
          v0 = fmaf(v0, v1, v2);
          v1 = fmaf(v1, v2, v0);
          v2 = fmaf(v2, v0, v1);
          v0 = fmaf(v0, v2, v1);
          if (g < 3) *reinterpret_cast<uint2*>(&lds[(tid + g * 256 + it) & 2047]) = make_uint2(__float_as_uint(v0), it);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        constexpr int TL[4][2] = {{1, 1}, {1, 0}, {2, 0}, {4, 0}};
#pragma unroll
        for (int y = 0; y < 4; ++y)
          accA[y][TL[g][0]] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[TL[g][0]]),
                                                                      __builtin_bit_cast(bf16x8, xb[3 * y + TL[g][1]]), accA[y][TL[g][0]], 0, 0, 0);
      }
    } else {
      // xb[6 c + 3 h + p] = piece p of column block c, K half h; wa[2 t + h] = tile t, K half h; p 0 -> tiles 0 1 2, p 1 -> 0 1, p 2 -> 0
#pragma unroll
      for (int g = 0; g < 6; ++g) {
        constexpr int TP[6][2] = {{0, 2}, {0, 1}, {1, 1}, {0, 0}, {1, 0}, {2, 0}};
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            accB[c][TP[g][0]] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa[2 * TP[g][0] + h]),
                                                                         __builtin_bit_cast(bf16x8, xb[6 * c + 3 * h + TP[g][1]]),
                                                                         accB[c][TP[g][0]], 0, 0, 0);
        if (MIX) {
          v0 = fmaf(v0, v1, v2);
          v1 = fmaf(v1, v2, v0);
          v2 = fmaf(v2, v0, v1);
          v0 = fmaf(v0, v2, v1);
          if (g < 3) *reinterpret_cast<uint2*>(&lds[(tid + g * 256 + it) & 2047]) = make_uint2(__float_as_uint(v0), it);
        }
      }
    }
    if (MIX) {
      st[0] = __float_as_uint(v0) + it;
      *reinterpret_cast<u32x4*>(out + ((size_t)(blockIdx.x * 256 + tid) * 4 + ((size_t)(it & 63) << 22))) = st;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MIX) {
#pragma unroll
      for (int q = 0; q < 12; ++q) xb[q] = xn[q];
#pragma unroll
      for (int q = 0; q < NW; ++q) wa[q] = wn[q];
    }
  }
  float s = v0 + v1 + v2;
#pragma unroll
  for (int y = 0; y < 4; ++y)
#pragma unroll
    for (int t = 0; t < 5; ++t) s += accA[y][t][0] + accA[y][t][3];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < 3; ++t) s += accB[c][t][0] + accB[c][t][15];
  out[blockIdx.x * 256 + tid] = s;
}

template <int SHAPE, int MIX>
void run(const char* name, float* out, u32x4* in, int iters) {
  const int wgs = 512;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, MIX>), dim3(wgs), dim3(256), 0, 0, out, in, iters);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, MIX>), dim3(wgs), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double steps = (double)wgs * 4 * iters;                    // wave K steps
  const double issued = steps * (SHAPE == 0 ? 40.0 * 2 * 16 * 16 * 32 : 24.0 * 2 * 32 * 32 * 16);
  const double useful = steps * 6.0 * 2 * 24 * 64 * 32;            // six products x 24 channels x 64 voxels x K 32
  printf("%-66s %8.3f ms  issued %5.0f TF  useful (6 x 24 x 64 x 32) %5.0f TF  %6.1f ns per wave K step\n", name, best, issued / best / 1e9,
         useful / best / 1e9, best * 1e6 / (iters * 1.0));
}

int main() {
  float* out;
  u32x4* in;
  hipMalloc(&out, ((size_t)64 << 22) * 4 + 512 * 256 * 16);
  hipMalloc(&in, 8192 * sizeof(u32x4));
  hipMemset(in, 0, 8192 * sizeof(u32x4));
  const int iters = 4000;
  run<0, 0>("A 16x16x32: 40 MFMAs per step, nothing else", out, in, iters);
  run<1, 0>("B 32x32x16: 24 MFMAs per step, nothing else", out, in, iters);
  run<0, 1>("A + 12 ds_read_b128 + 5 L2 loads + 24 VALU + 3 ds_write_b64 + 1 store", out, in, iters);
  run<1, 1>("B + 12 ds_read_b128 + 6 L2 loads + 24 VALU + 3 ds_write_b64 + 1 store", out, in, iters);
  return 0;
}
