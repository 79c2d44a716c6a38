// v_mfma_f32_16x16x32_bf16: cycles per MFMA of one SIMD as a function of the dependency distance (D accumulators in rotation) and of
// the waves per SIMD (1 or 2).   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_dep.hip -o tools/ubench/mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int D>
__global__ __launch_bounds__(512, 1) void k(float* out, long long* cyc, int iters) {
  f32x4 acc[D];
#pragma unroll
  for (int i = 0; i < D; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < D; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < D; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int D>
void run(int threads) {
  float* o; long long* c;
  hipMalloc(&o, 512 * 4); hipMalloc(&c, 8);
  const int iters = 2000;
  hipLaunchKernelGGL(k<D>, dim3(1), dim3(threads), 0, 0, o, c, iters);
  hipLaunchKernelGGL(k<D>, dim3(1), dim3(threads), 0, 0, o, c, iters);
  long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  const int waves_per_simd = threads / 256;
  printf("distance %d, %d wave(s) per SIMD: %.1f cycles per MFMA of a wave, %.1f per MFMA of the SIMD\n", D, waves_per_simd,
         (double)h / (iters * 8.0 * D), (double)h / (iters * 8.0 * D * waves_per_simd));
  hipFree(o); hipFree(c);
}
// the whole chip (256 workgroups), wall clock next to the tick counter: what a tick is, and the chip's MFMA rate
template <int D>
void chip(int threads, int nwg = 256) {
  float* o; long long* c;
  hipMalloc(&o, (size_t)nwg * 512 * 4); hipMalloc(&c, nwg * 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<D>, dim3(nwg), dim3(threads), 0, 0, o, c, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<D>, dim3(nwg), dim3(threads), 0, 0, o, c, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  const double mfmas = (double)nwg * (threads / 64) * iters * 8.0 * D;
  printf("chip: %d workgroups x %d threads, distance %d: %.3f ms, %lld ticks (%.2f G ticks/s), %.0f TFLOP/s, %.2f ticks per MFMA of a SIMD\n",
         nwg, threads, D, ms, h, h / (ms * 1e6), mfmas * 16384 / (ms * 1e-3) / 1e12, (double)h / (iters * 8.0 * D * (threads / 256)));
  hipFree(o); hipFree(c);
}
int main() {
  run<1>(256); run<2>(256); run<3>(256); run<4>(256);
  run<1>(512); run<2>(512); run<3>(512); run<4>(512);
  chip<4>(256); chip<4>(512); chip<4>(256);
  chip<4>(256, 1); chip<4>(512, 1); chip<4>(256, 1); chip<4>(512, 8); chip<4>(512, 64);
  return 0;
}
