// regression_metric='ssim' of SynthSR/metrics_model.py:105-125: minus the mean structural similarity of predicted and
// target volume, evaluated slice-wise with tf.image.ssim(max_val=1): 11x11 Gaussian window (sigma 1.5, 'VALID'),
// k1 = .01, k2 = .03 (TensorFlow 2.0 image_ops_impl.py `_ssim_per_channel` / `_ssim_helper` / `_fspecial_gauss`;
// TensorFlow is a third-party dependency of the reference, pinned to 2.0.0 in requirements.txt, not vendored):
//   mu_x = w*x, mu_y = w*y, e_xy = w*(xy), e_2 = w*(x^2+y^2)        (w: normalised 2-D Gaussian = outer product of 1-D)
//   lum = (2 mu_x mu_y + c1) / (mu_x^2 + mu_y^2 + c1),  cs = (2 e_xy - 2 mu_x mu_y + c2) / (e_2 - mu_x^2 - mu_y^2 + c2)
//   ssim = mean over window positions and slices of lum * cs
// The reference sums three such terms with weight -1/3: slices across axis 0 windows over (1,2); the same volume with axes
// 1 and 2 swapped (windows over (2,1): the SAME value, the window being symmetric); slices across axis 1, windows (2,0).
// Everything is HBM-bound elementwise / 11-tap stencil work on 16 MB maps: separable 1-D passes, planar maps.
#include "common.h"

SYN_DET_SETTER(ssim)

namespace {

constexpr int SSIM_TAPS = 11;
struct SsimTaps {
  float w[SSIM_TAPS];
};
struct SsimBox {
  int d1, d2;      // full volume's trailing sizes
  int lo[3], n[3];  // box origin and size
};

// maps[0..3] = x, y, x*y, x*x + y*y over the box (x = pred, y = target), planar [4][n0*n1*n2]
__global__ __launch_bounds__(256) void ssim_products_kernel(const float* __restrict__ pred,
                                                            const float* __restrict__ target, SsimBox b,
                                                            float* __restrict__ maps) {
  const int64_t nb = (int64_t)b.n[0] * b.n[1] * b.n[2];
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < nb; i += (int64_t)gridDim.x * 256) {
    const int x2 = (int)(i % b.n[2]), x1 = (int)((i / b.n[2]) % b.n[1]), x0 = (int)(i / ((int64_t)b.n[1] * b.n[2]));
    const int64_t v = ((int64_t)(x0 + b.lo[0]) * b.d1 + (x1 + b.lo[1])) * b.d2 + (x2 + b.lo[2]);
    const float x = pred[v], y = target[v];
    maps[i] = x;
    maps[nb + i] = y;
    maps[2 * nb + i] = x * y;
    maps[3 * nb + i] = x * x + y * y;
  }
}

// 11-tap correlation along `axis` of nmaps planar volumes [s0][s1][s2]: valid (out length s - 10) or full (s + 10, the
// transpose of valid: out[i] = sum_t w[t] in[i + t - 10], zero outside)
__global__ __launch_bounds__(256) void ssim_filter_kernel(const float* __restrict__ in, float* __restrict__ out, int s0,
                                                          int s1, int s2, int axis, int full, int nmaps, SsimTaps taps) {
  int o[3] = {s0, s1, s2};
  o[axis] += full ? (SSIM_TAPS - 1) : -(SSIM_TAPS - 1);
  const int64_t nin = (int64_t)s0 * s1 * s2, nout = (int64_t)o[0] * o[1] * o[2];
  const int64_t stride = axis == 2 ? 1 : (axis == 1 ? s2 : (int64_t)s1 * s2);
  const int len = axis == 0 ? s0 : (axis == 1 ? s1 : s2);
  for (int64_t j = blockIdx.x * (int64_t)256 + threadIdx.x; j < nout * nmaps; j += (int64_t)gridDim.x * 256) {
    const int m = (int)(j / nout);
    const int64_t i = j - (int64_t)m * nout;
    int c[3] = {(int)(i / ((int64_t)o[1] * o[2])), (int)((i / o[2]) % o[1]), (int)(i % o[2])};
    const int p = c[axis] - (full ? (SSIM_TAPS - 1) : 0);  // first input index along the axis
    c[axis] = 0;
    const float* src = in + (int64_t)m * nin + ((int64_t)c[0] * s1 + c[1]) * s2 + c[2];
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < SSIM_TAPS; ++t) {
      const int q = p + t;
      if (q >= 0 && q < len) acc += taps.w[t] * src[(int64_t)q * stride];
    }
    out[j] = acc;
  }
}

// f[0..3] = filtered maps (mu_x, mu_y, e_xy, e_2) on the nq window positions.  loss += scale * sum lum*cs;
// g[0..2] = scale * dS/d(mu_x, e_xy, e_2)
__global__ __launch_bounds__(256) void ssim_point_kernel(const float* __restrict__ f, int64_t nq, float c1, float c2,
                                                         float scale, float* __restrict__ loss, float* __restrict__ g) {
  float lsum = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < nq; i += (int64_t)gridDim.x * 256) {
    const float mx = f[i], my = f[nq + i], exy = f[2 * nq + i], e2 = f[3 * nq + i];
    const float n0 = 2.f * mx * my, d0 = mx * mx + my * my;
    const float N0 = n0 + c1, D0 = d0 + c1, N1 = 2.f * exy - n0 + c2, D1 = e2 - d0 + c2;
    const float iD0 = 1.f / D0, iD1 = 1.f / D1;
    const float lum = N0 * iD0, cs = N1 * iD1;
    lsum += lum * cs;
    if (g) {
      const float dlum = (2.f * my - lum * 2.f * mx) * iD0;
      const float dcs = (-2.f * my + cs * 2.f * mx) * iD1;
      g[i] = scale * (dlum * cs + lum * dcs);
      g[nq + i] = scale * lum * 2.f * iD1;
      g[2 * nq + i] = -scale * lum * cs * iD1;
    }
  }
  lsum = syn_wave_sum(lsum);
  __shared__ float wsum[4];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) wsum[0] = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * scale;
  __syncthreads();
  if (syn_det_gather(wsum, 1))
    if (threadIdx.x == 0) atomicAdd(loss, wsum[0]);
  syn_det_gather_end(1);
}

// dpred[v in box] += GA + y GB + 2 x GC  (GA, GB, GC: the three gradient maps filtered back onto the box)
__global__ __launch_bounds__(256) void ssim_combine_kernel(const float* __restrict__ gb, const float* __restrict__ pred,
                                                           const float* __restrict__ target, SsimBox b,
                                                           float* __restrict__ dpred) {
  const int64_t nb = (int64_t)b.n[0] * b.n[1] * b.n[2];
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < nb; i += (int64_t)gridDim.x * 256) {
    const int x2 = (int)(i % b.n[2]), x1 = (int)((i / b.n[2]) % b.n[1]), x0 = (int)(i / ((int64_t)b.n[1] * b.n[2]));
    const int64_t v = ((int64_t)(x0 + b.lo[0]) * b.d1 + (x1 + b.lo[1])) * b.d2 + (x2 + b.lo[2]);
    dpred[v] += gb[i] + target[v] * gb[nb + i] + 2.f * pred[v] * gb[2 * nb + i];
  }
}

bool make_box(const int* shape, const int* crop, SsimBox& b) {
  if (!shape || shape[0] < 1 || shape[1] < 1 || shape[2] < 1) return false;
  b.d1 = shape[1];
  b.d2 = shape[2];
  for (int i = 0; i < 3; ++i) {
    b.lo[i] = crop ? crop[i] : 0;
    b.n[i] = crop ? crop[3 + i] : shape[i];
    if (b.lo[i] < 0 || b.n[i] < 1 || b.lo[i] + b.n[i] > shape[i]) return false;
  }
  return true;
}

}  // namespace

extern "C" {

int synthsr_ssim_products(const float* pred, const float* target, const int* shape, const int* crop, float* maps,
                          synthsr_stream_t stream) {
  SsimBox b;
  if (!pred || !target || !maps || !make_box(shape, crop, b)) return SYNTHSR_EINVAL;
  const int64_t nb = (int64_t)b.n[0] * b.n[1] * b.n[2];
  hipLaunchKernelGGL(ssim_products_kernel, dim3(syn_grid(nb, 256)), dim3(256), 0, (hipStream_t)stream, pred, target, b,
                     maps);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_ssim_filter(const float* in, float* out, const int* shape, int axis, int full, int nmaps, const float* taps,
                        synthsr_stream_t stream) {
  if (!in || !out || !shape || !taps || axis < 0 || axis > 2 || nmaps < 1) return SYNTHSR_EINVAL;
  if (shape[0] < 1 || shape[1] < 1 || shape[2] < 1 || (!full && shape[axis] < SSIM_TAPS)) return SYNTHSR_EINVAL;
  SsimTaps t;
  for (int i = 0; i < SSIM_TAPS; ++i) t.w[i] = taps[i];
  int o[3] = {shape[0], shape[1], shape[2]};
  o[axis] += full ? (SSIM_TAPS - 1) : -(SSIM_TAPS - 1);
  const int64_t nout = (int64_t)o[0] * o[1] * o[2] * nmaps;
  hipLaunchKernelGGL(ssim_filter_kernel, dim3(syn_grid(nout, 256)), dim3(256), 0, (hipStream_t)stream, in, out, shape[0],
                     shape[1], shape[2], axis, full ? 1 : 0, nmaps, t);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_ssim_point(const float* filtered, int64_t nq, float max_val, float scale, float* loss, float* grads,
                       synthsr_stream_t stream) {
  if (!filtered || !loss || nq < 1) return SYNTHSR_EINVAL;
  const float c1 = (0.01f * max_val) * (0.01f * max_val), c2 = (0.03f * max_val) * (0.03f * max_val);
  hipLaunchKernelGGL(ssim_point_kernel, dim3(syn_grid(nq, 256, 1024)), dim3(256), 0, (hipStream_t)stream, filtered, nq, c1,
                     c2, scale, loss, grads);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

int synthsr_ssim_combine(const float* gback, const float* pred, const float* target, const int* shape, const int* crop,
                         float* dpred, synthsr_stream_t stream) {
  SsimBox b;
  if (!gback || !pred || !target || !dpred || !make_box(shape, crop, b)) return SYNTHSR_EINVAL;
  const int64_t nb = (int64_t)b.n[0] * b.n[1] * b.n[2];
  hipLaunchKernelGGL(ssim_combine_kernel, dim3(syn_grid(nb, 256)), dim3(256), 0, (hipStream_t)stream, gback, pred, target,
                     b, dpred);
  SYN_CHECK_LAUNCH();
  return SYNTHSR_OK;
}

}  // extern "C"
