// Probe of ds_read_b64_tr_b16 (gfx950): which LDS element lands in which lane / register slot.
// LDS holds u16 value = its own element index.  Lane l supplies the byte address addr[l]; out[l*4 + j] = the j-th
// 16-bit value it receives.   hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe && ./tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(const int* __restrict__ addr, uint16_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const uint32_t a = (uint32_t)(uintptr_t)lds + (uint32_t)addr[threadIdx.x];
  uint64_t r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)(r >> (16 * j));
}

static void run(const char* name, const std::vector<int>& addr) {
  int* d_a;
  uint16_t* d_o;
  hipMalloc(&d_a, 64 * 4);
  hipMalloc(&d_o, 64 * 4 * 2);
  hipMemcpy(d_a, addr.data(), 64 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_a, d_o);
  std::vector<uint16_t> o(256);
  hipMemcpy(o.data(), d_o, 512, hipMemcpyDeviceToHost);
  printf("== %s\n", name);
  for (int l = 0; l < 64; ++l)
    printf("lane %2d addr(elem) %4d -> %4d %4d %4d %4d\n", l, addr[l] / 2, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
  hipFree(d_a);
  hipFree(d_o);
}

int main() {
  std::vector<int> a(64);
  // P1: every lane its own 8-byte chunk, linear: lane l -> elements 4l..4l+3
  for (int l = 0; l < 64; ++l) a[l] = 8 * l;
  run("linear 8B per lane", a);
  // P2: per 16-lane group a [4][16] row-major block with a padded row stride of 40 elements:
  //     lane l -> row (l%16)/4, col chunk (l%4), group (l/16) at 1000*group
  for (int l = 0; l < 64; ++l) a[l] = 2 * (1000 * (l / 16) + ((l % 16) / 4) * 40 + (l % 4) * 4);
  run("row (l%16)/4 stride 40, chunk l%4", a);
  // P3: lane l -> row l%4, chunk (l%16)/4
  for (int l = 0; l < 64; ++l) a[l] = 2 * (1000 * (l / 16) + (l % 4) * 40 + ((l % 16) / 4) * 4);
  run("row l%4 stride 40, chunk (l%16)/4", a);
  return 0;
}
