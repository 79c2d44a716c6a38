// micro-benchmark: what sustains the fp32 MFMA issue rate next to LDS reads / VALU copies (2 waves per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int VK = 0, int VN = 0>
__global__ __launch_bounds__(256, 2) void k(float* out, const float* in, int iters) {
  __shared__ float lds[8192];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += 256) lds[i] = in[i];
  __syncthreads();
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float2 a[4], an[4];
  int r[8];
  int sreg = iters;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = tid + i;
  float2 b[2] = {make_float2(in[tid], in[tid + 1]), make_float2(in[tid + 2], in[tid + 3])};
  int base = (tid & 63) * 28 + (tid >> 6) * 8;
#pragma unroll
  for (int m = 0; m < 4; ++m) a[m] = *reinterpret_cast<const float2*>(&lds[(base + m * 504) & 8190]);
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 1) {
#pragma unroll
      for (int m = 0; m < 4; ++m) an[m] = *reinterpret_cast<const float2*>(&lds[(base + m * 504 + it * 8) & 8190]);
    }
    if (MODE >= 3) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].x, b[n].x, acc[m * 2 + n], 0, 0, 0);
    // VN independent VALU instructions of kind VK between the two MFMA batches
#pragma unroll
    for (int v = 0; v < VN; ++v) {
      if (VK == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[v & 7]) : "v"(tid));
      if (VK == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[v & 7]) : "v"(tid));
      if (VK == 3) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sreg));
      if (VK == 4) asm volatile("s_nop 0");
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].y, b[n].y, acc[m * 2 + n], 0, 0, 0);
    if (MODE >= 3) __builtin_amdgcn_sched_barrier(0);
    if (MODE >= 2) {
#pragma unroll
      for (int m = 0; m < 4; ++m) a[m] = an[m];
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (float)r[i];
  s += (float)sreg;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int VK = 0, int VN = 0>
void run(const char* name, float* out, float* in, int wgs, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, VK, VN>), dim3(wgs), dim3(256), 0, 0, out, in, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, VK, VN>), dim3(wgs), dim3(256), 0, 0, out, in, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)wgs * 4 * iters * 16 * 2048.0;
  printf("%-40s wgs=%d %.3f ms  %.1f TF\n", name, wgs, ms, fl / ms / 1e9);
}

int main() {
  float *out, *in;
  hipMalloc(&out, 1 << 24);
  hipMalloc(&in, 1 << 20);
  hipMemset(in, 0, 1 << 20);
  for (int wgs : {512, 4096}) {
    int iters = wgs == 512 ? 81 * 32 : 81 * 4;
    run<0>("mfma only", out, in, wgs, iters);
    run<1>("+ 4 ds_read_b64 per 16 mfma", out, in, wgs, iters);
    run<2>("+ reads + 8 v_mov copies", out, in, wgs, iters);
    run<3>("+ reads + copies + sched_barrier", out, in, wgs, iters);
    run<1, 1, 8>("reads + 8 v_add_u32 (independent)", out, in, wgs, iters);
    run<1, 1, 16>("reads + 16 v_add_u32", out, in, wgs, iters);
    run<1, 1, 32>("reads + 32 v_add_u32", out, in, wgs, iters);
    run<1, 1, 64>("reads + 64 v_add_u32", out, in, wgs, iters);
    run<1, 2, 8>("reads + 8 v_mul_lo_u32", out, in, wgs, iters);
    run<1, 2, 16>("reads + 16 v_mul_lo_u32", out, in, wgs, iters);
    run<1, 3, 32>("reads + 32 s_add_u32", out, in, wgs, iters);
    run<1, 4, 32>("reads + 32 s_nop", out, in, wgs, iters);
  }
  return 0;
}
