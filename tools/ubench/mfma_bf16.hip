// micro-benchmark: sustained rate of v_mfma_f32_16x16x32_bf16 (the bf16 conv kernels' instruction), MFMA only and next to the
// operand traffic of one K-step of conv3d_bf16_fwd_kernel (6 ds_read_b128 per 8 MFMAs), 1-3 waves per SIMD
//   hipcc --offload-arch=gfx950 -O3 -o mfma_bf16 mfma_bf16.hip && ./mfma_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int OCC>
__global__ __launch_bounds__(256, OCC) void k(float* out, const uint4* in, int iters) {
  __shared__ uint4 lds[2048];
  const int tid = threadIdx.x;
  for (int i = tid; i < 2048; i += 256) lds[i] = in[i];
  __syncthreads();
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a[2], b[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const bf16x8*>(&lds[(tid + 64 * i) & 2047]);
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const bf16x8*>(&lds[(tid * 3 + 128 * i) & 2047]);
  for (int it = 0; it < iters; ++it) {
    bf16x8 an[2], bn[4];
    if (MODE >= 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i) an[i] = *reinterpret_cast<const bf16x8*>(&lds[(tid + 64 * i + it) & 2047]);
#pragma unroll
      for (int i = 0; i < 4; ++i) bn[i] = *reinterpret_cast<const bf16x8*>(&lds[(tid * 3 + 128 * i + it) & 2047]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[n * 2 + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[n * 2 + m], 0, 0, 0);
    if (MODE >= 1) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = an[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = bn[i];
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int OCC>
void run(const char* name, float* out, uint4* in, int iters) {
  const int wgs = 256 * OCC;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, OCC>), dim3(wgs), dim3(256), 0, 0, out, in, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, OCC>), dim3(wgs), dim3(256), 0, 0, out, in, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)wgs * 4 * iters * 8 * (2.0 * 16 * 16 * 32);
  printf("%-52s %d waves/SIMD  %.3f ms  %.0f TFLOP/s\n", name, OCC, ms, fl / ms / 1e9);
}

int main() {
  float* out;
  uint4* in;
  hipMalloc(&out, 256 * 3 * 256 * sizeof(float));
  hipMalloc(&in, 2048 * sizeof(uint4));
  hipMemset(in, 0, 2048 * sizeof(uint4));
  const int iters = 20000;
  run<0, 1>("MFMA 16x16x32 bf16 only", out, in, iters);
  run<0, 2>("MFMA 16x16x32 bf16 only", out, in, iters);
  run<1, 1>("+ 6 ds_read_b128 per 8 MFMAs (conv K-step)", out, in, iters);
  run<1, 2>("+ 6 ds_read_b128 per 8 MFMAs (conv K-step)", out, in, iters);
  run<1, 3>("+ 6 ds_read_b128 per 8 MFMAs (conv K-step)", out, in, iters);
  return 0;
}
