// Probe of the gfx950 LDS-DMA buffer load (buffer_load_dwordx4 ... offen lds, via __builtin_amdgcn_raw_ptr_buffer_load_lds):
//  (1) destination = wave-uniform LDS base + lane * 16;  (2) what lands in LDS for a lane whose offset is out of range
//  (>= num_records): zeros (usable as conv zero padding) or nothing (sentinel survives);  (3) completion is visible after
//  s_waitcnt vmcnt(0) + barrier.      hipcc --offload-arch=gfx950 -O3 tools/ubench/glds_probe.hip -o tools/ubench/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const float* in, int nbytes, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[256 * 4];
  for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = -777.f;  // sentinel
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, nbytes, 0x00020000);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // even lanes read in range, odd lanes out of range (offset 0x80000000)
  const unsigned off = (lane & 1) ? 0x80000000u : (unsigned)(threadIdx.x * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + wave * 256), 16, (int)off, 0, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256) out[i] = lds[i];
}
int main() {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  float *din, *dout;
  hipMalloc(&din, 4096);
  hipMalloc(&dout, 4096);
  hipMemcpy(din, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, din, 4096, dout);
  std::vector<float> o(1024);
  hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
  int ok_in = 0, zero_oob = 0, sentinel_oob = 0, other = 0;
  for (int t = 0; t < 256; ++t)
    for (int j = 0; j < 4; ++j) {
      const float v = o[t * 4 + j];
      if (t & 1) {
        if (v == 0.f) ++zero_oob; else if (v == -777.f) ++sentinel_oob; else ++other;
      } else if (v == (float)(t * 4 + j)) ++ok_in; else ++other;
    }
  printf("in-range lanes correct: %d / 512; out-of-range lanes: %d zeros, %d sentinels untouched, %d other\n", ok_in, zero_oob,
         sentinel_oob, other);
  printf("first values: %g %g %g %g | %g %g %g %g\n", o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
  return 0;
}
