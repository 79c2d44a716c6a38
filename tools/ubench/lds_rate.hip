// LDS read throughput of one CU (8 waves of one workgroup): ds_read_b64 vs ds_read_b64_tr_b16 vs ds_read_b128, conflict-free
// linear addresses (lane l -> its own 8 / 16 bytes), and the weight-gradient kernels' transposed pattern (x image rows of 32 B).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_rate.hip -o tools/ubench/lds_rate && tools/ubench/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(512, 1) void probe(long long* out, int iters, int pat) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 512) reinterpret_cast<uint32_t*>(lds)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t a;
  if (pat == 0) a = (uint32_t)(wave * 8192 + lane * (MODE == 2 ? 16 : 8));
  else {  // transposed-read pattern of the weight-gradient kernels: 16-lane group g: voxel 4 (g & 1) + (li >> 2) of row g >> 1, quad li & 3
    const int g = lane >> 4, li = lane & 15;
    a = (uint32_t)(wave * 8192 + (((g >> 1) * 18) + 4 * (g & 1) + (li >> 2)) * 32 + (li & 3) * 8);
  }
  a += (uint32_t)(uintptr_t)lds;
  uint64_t acc0 = 0, acc1 = 0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if constexpr (MODE == 0) {
        uint64_t r;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(u * 512));
        acc0 ^= r;
      } else if constexpr (MODE == 1) {
        uint64_t r;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(u * 512));
        acc0 ^= r;
      } else {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        u32x4 r;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(u * 1024));
        acc0 ^= r[0] ^ r[2];
        acc1 ^= r[1] ^ r[3];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const long long t1 = clock64();
  if (lane == 0) out[wave] = t1 - t0;
  if (acc0 == 0x1234567 && acc1 == 77) out[8] = 1;
}

int main() {
  long long* d;
  hipMalloc(&d, 16 * 8);
  const int iters = 20000;
  for (int pat = 0; pat < 2; ++pat)
    for (int mode = 0; mode < 3; ++mode) {
      if (pat == 1 && mode == 2) continue;
      auto k = mode == 0 ? probe<0> : (mode == 1 ? probe<1> : probe<2>);
      hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(k, dim3(1), dim3(512), 160 * 1024, 0, d, iters, pat);
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(1), dim3(512), 160 * 1024, 0, d, iters, pat);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      long long h[8];
      hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
      long long mx = 0;
      for (int w = 0; w < 8; ++w) mx = h[w] > mx ? h[w] : mx;
      const double bytes = 8.0 * iters * 8 * 64 * (mode == 2 ? 16 : 8);
      printf("%-20s %-10s: %lld ticks, %.3f ms (launch incl. fill of 160 KB) for %d x 8 reads x 8 waves = %.1f bytes / tick / CU, %.0f GB/s "
             "of one CU = %.1f bytes / cycle at 2.4 GHz\n",
             mode == 0 ? "ds_read_b64" : (mode == 1 ? "ds_read_b64_tr_b16" : "ds_read_b128"), pat ? "wgrad-tr" : "linear", mx, ms, iters,
             bytes / mx, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 2.4e9);
    }
  return 0;
}
