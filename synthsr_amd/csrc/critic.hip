// Pointwise / dense pieces of the WGAN-GP critic of SynthSR/fine_tuning_with_adversary.py:482-508 (`make_discriminator`:
// [Conv3D(3, stride 1) + LeakyReLU(.2), Conv3D(3, stride 2) + LeakyReLU(.2)] x n_levels -> Flatten -> Dense ->
// LeakyReLU(.2) -> Dense(1)).  The 3x3x3 convolutions, strided ones included, run through conv3d.hip (ops.conv3d_stride2*).
// All HBM-bound.
#include "common.h"

SYN_DET_SETTER(critic)

namespace {

// y = x > 0 ? x : alpha x (in place allowed); with dy: dx = dy * (y > 0 ? 1 : alpha), y being the layer OUTPUT (its sign is
// the input's sign)
__global__ __launch_bounds__(256) void leaky_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                    float* __restrict__ out, int64_t n, float alpha) {
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    out[i] = dy ? dy[i] * (v > 0.f ? 1.f : alpha) : (v > 0.f ? v : alpha * v);
  }
}

// out = leaky(x + bias[c]) for channels-last x [n / C][C] (in place allowed)
__global__ __launch_bounds__(256) void bias_leaky_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                         float* __restrict__ out, int64_t n, int C, float alpha) {
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i] + bias[(int)(i % C)];
    out[i] = v > 0.f ? v : alpha * v;
  }
}

// out[c] += sum over rows of x [n][C]: threads run along the channels, 256 / C rows in flight per workgroup
__global__ __launch_bounds__(256) void colsum_rows_kernel(const float* __restrict__ x, int64_t n, int C,
                                                          float* __restrict__ out) {
  __shared__ float part[256];
  const int rows = C <= 256 ? 256 / C : 1;
  int* turn = nullptr;
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int width = min(256, C - c0);
    const int r = threadIdx.x / width, c = threadIdx.x - r * width;
    float acc = 0.f;
    if (r < rows)
      for (int64_t v = (int64_t)blockIdx.x * rows + r; v < n; v += (int64_t)gridDim.x * rows) acc += x[v * C + c0 + c];
    part[threadIdx.x] = acc;
    __syncthreads();
    if (c0 == 0) turn = syn_turn_begin(0, blockIdx.x);
    if (threadIdx.x < width) {
      float t = 0.f;
      for (int k = 0; k < rows; ++k) t += part[k * width + threadIdx.x];
      atomicAdd(out + c0 + threadIdx.x, t);
    }
    __syncthreads();
  }
  syn_turn_end(turn, blockIdx.x, gridDim.x);
}

// Dense layer y[j] = b[j] + sum_i x[i] W[i][j]  (W [n_in][n_out], Keras layout).  One workgroup per slab of rows, partial
// sums to LDS, then atomics on the n_out outputs (n_out <= 1024): W (up to 524 MB) is streamed once, coalesced along j.
__global__ __launch_bounds__(256) void dense_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                        const float* __restrict__ b, float* __restrict__ y, int64_t n_in,
                                                        int n_out, int rows_per_block) {
  extern __shared__ float acc[];  // [n_out]
  for (int j = threadIdx.x; j < n_out; j += 256) acc[j] = (blockIdx.x == 0 && b) ? b[j] : 0.f;
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(n_in, r0 + rows_per_block);
  for (int j = threadIdx.x; j < n_out; j += 256) {
    float s = 0.f;
    for (int64_t i = r0; i < r1; ++i) s += x[i] * W[i * n_out + j];
    acc[j] += s;
  }
  __syncthreads();
  if (syn_det_gather(acc, n_out))
    for (int j = threadIdx.x; j < n_out; j += 256) atomicAdd(&y[j], acc[j]);
  syn_det_gather_end(n_out);
}

// backward of the dense layer: dx[i] = sum_j W[i][j] dy[j] (optional), dW[i][j] += scale_w * x[i] dy[j] (optional)
__global__ __launch_bounds__(256) void dense_bwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                        const float* __restrict__ dy, float* __restrict__ dx,
                                                        float* __restrict__ dW, int64_t n_in, int n_out) {
  extern __shared__ float sdy[];  // [n_out]
  for (int j = threadIdx.x; j < n_out; j += 256) sdy[j] = dy[j];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < n_in; i += (int64_t)gridDim.x * 4) {  // one wave per row
    const float xi = x ? x[i] : 0.f;
    float s = 0.f;
    for (int j = lane; j < n_out; j += 64) {
      if (dx) s += W[i * n_out + j] * sdy[j];
      if (dW) dW[i * n_out + j] += xi * sdy[j];
    }
    if (dx) {
      s = syn_wave_sum(s);
      if (lane == 0) dx[i] = s;
    }
  }
}

// out = x * y (elementwise); with lut != NULL: out[i] = lut[labels[i]] (0 for labels outside [0, n_lut)) - ConvertLabels
__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                  float* __restrict__ out, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = x[i] * y[i];
}
__global__ __launch_bounds__(256) void lut_kernel(const int* __restrict__ labels, const float* __restrict__ lut, int n_lut,
                                                  float* __restrict__ out, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int l = labels[i];
    out[i] = ((unsigned)l < (unsigned)n_lut) ? lut[l] : 0.f;
  }
}

// out = a * x + b * y (elementwise; y optional)
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                    float* __restrict__ out, int64_t n, float a, float b) {
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}

// *out += sum x^2
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  float s = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += x[i] * x[i];
  s = syn_wave_sum(s);
  __shared__ float w[4];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) w[0] = w[0] + w[1] + w[2] + w[3];
  __syncthreads();
  if (syn_det_gather(w, 1))
    if (threadIdx.x == 0) atomicAdd(out, w[0]);
  syn_det_gather_end(1);
}

}  // namespace

extern "C" {

int synthsr_leaky_relu(const float* x, const float* dy, float* out, int64_t n, float alpha, synthsr_stream_t stream) {
  if (!x || !out || n < 1) return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(leaky_kernel, dim3(syn_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, out, n, alpha);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_bias_leaky_relu(const float* x, const float* bias, float* out, int64_t n, int C, float alpha,
                            synthsr_stream_t stream) {
  if (!x || !bias || !out || n < 1 || C < 1) return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(bias_leaky_kernel, dim3(syn_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, bias, out, n, C,
                     alpha);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_colsum(const float* x, int64_t n, int C, float* out, synthsr_stream_t stream) {
  if (!x || !out || n < 1 || C < 1) return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(colsum_rows_kernel, dim3(syn_grid(n, 4, 1024)), dim3(256), 0, (hipStream_t)stream, x, n, C, out);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_dense_fwd(const float* x, const float* W, const float* b, float* y, int64_t n_in, int n_out,
                      synthsr_stream_t stream) {
  if (!x || !W || !y || n_in < 1 || n_out < 1 || n_out > 1024) return SYNTHSR_EINVAL;
  if (hipMemsetAsync(y, 0, (size_t)n_out * sizeof(float), (hipStream_t)stream) != hipSuccess) return SYNTHSR_ELAUNCH;
  const int rows = (int)std::max<int64_t>(1, syn_cdiv(n_in, 2048));
  hipLaunchKernelGGL(dense_fwd_kernel, dim3((unsigned)syn_cdiv(n_in, rows)), dim3(256), n_out * sizeof(float),
                     (hipStream_t)stream, x, W, b, y, n_in, n_out, rows);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_dense_bwd(const float* x, const float* W, const float* dy, float* dx, float* dW, int64_t n_in, int n_out,
                      synthsr_stream_t stream) {
  if (!W || !dy || (!dx && !dW) || (dW && !x) || n_in < 1 || n_out < 1 || n_out > 1024) return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(dense_bwd_kernel, dim3(syn_grid(n_in, 4, 4096)), dim3(256), n_out * sizeof(float), (hipStream_t)stream,
                     x, W, dy, dx, dW, n_in, n_out);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_mul(const float* x, const float* y, float* out, int64_t n, synthsr_stream_t stream) {
  if (!x || !y || !out || n < 1) return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(mul_kernel, dim3(syn_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, out, n);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_lut_gather(const int* labels, const float* lut, int n_lut, float* out, int64_t n, synthsr_stream_t stream) {
  if (!labels || !lut || !out || n < 1 || n_lut < 1) return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(lut_kernel, dim3(syn_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, labels, lut, n_lut, out, n);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_axpby(const float* x, const float* y, float* out, int64_t n, float a, float b, synthsr_stream_t stream) {
  if (!x || !out || n < 1) return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(axpby_kernel, dim3(syn_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, out, n, a, b);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_sumsq(const float* x, int64_t n, float* out, synthsr_stream_t stream) {
  if (!x || !out || n < 1) return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(sumsq_kernel, dim3(syn_grid(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, x, n, out);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

}  // extern "C"
