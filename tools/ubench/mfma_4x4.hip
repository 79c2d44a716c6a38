// micro-benchmark: issue rate of v_mfma_f32_4x4x1_16B_f32 (16 blocks of 4x4 outer products, A broadcast via CBSZ/ABID)
// compared with v_mfma_f32_16x16x4_f32.  Flops: 4x4x1 = 512 / instr, 16x16x4 = 2048 / instr.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, const float* in, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += 256) lds[i] = in[i];
  __syncthreads();
  f32x4 acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float w[3] = {in[tid], in[tid + 256], in[tid + 512]};
  const int base = (tid & 63) * 28 + (tid >> 6) * 2048;
  float4 x = *reinterpret_cast<const float4*>(&lds[base & 8188]);
  for (int it = 0; it < iters; ++it) {
    float4 xn = x;
    if (MODE >= 1) xn = *reinterpret_cast<const float4*>(&lds[(base + (it & 3) * 4) & 8188]);
    // 4 k's x 6 channel groups = 24 MFMAs; weights of 8 k's live in w[0..2] (48 groups), group = abid
    const float xs[4] = {x.x, x.y, x.z, x.w};
    sfor<0, 24>([&](auto GI) {
      constexpr int gi = decltype(GI)::value, kk = gi / 6, g = gi % 6;
      if (MODE == 2)  // two voxel sets share the weights: 12 accumulators
        acc[6 + g] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[(gi + 24) / 16], xs[kk], acc[6 + g], 4, (gi + 24) % 16, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[gi / 16], xs[kk], acc[g], 4, gi % 16, 0);
    });
    x = xn;
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, float* out, float* in, int wgs, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, in, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, in, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double per_it = (MODE == 2 ? 48 : 24) * 512.0;
  double fl = (double)wgs * 4 * iters * per_it;
  printf("%-44s wgs=%d %.3f ms  %.1f TF\n", name, wgs, ms, fl / ms / 1e9);
}

int main() {
  float *out, *in;
  hipMalloc(&out, 1 << 24);
  hipMalloc(&in, 1 << 20);
  hipMemset(in, 0, 1 << 20);
  for (int wgs : {256, 512, 1024}) {
    int iters = 4096;
    run<0>("4x4x1 cbsz=4: 24 mfma / iter", out, in, wgs, iters);
    run<1>("4x4x1 + ds_read_b128 per 24 mfma", out, in, wgs, iters);
    run<2>("4x4x1 two voxel sets (48 mfma) + read", out, in, wgs, iters / 2);
  }
  return 0;
}
