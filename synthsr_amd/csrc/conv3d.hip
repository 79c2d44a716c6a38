// 3x3x3 'same' Conv3D for NDHWC float32 on gfx950 as implicit GEMM on the exact-f32 matrix cores
// (v_mfma_f32_16x16x4_f32: 64 lanes, A[16 x 4], B[4 x 16], D[16 x 16] in 4 accumulator registers).
//
//   forward / data-gradient:  D[voxel][co] += A[voxel][(tap,ci)] * B[(tap,ci)][co]
//       workgroup = 4 waves = 4x4x16 output voxels; the 6x6x18 halo tile of one 24-channel chunk is
//       staged once in LDS ([voxel][24+4 pad] -> conflict-free ds_read_b64 of 2 channels / lane) and
//       reused by all 27 taps; the B fragments are pre-packed in fragment order (one 512 B coalesced
//       load per wave and fragment, served from L2) and double-buffered in registers one tap ahead.
//   weight-gradient:          D[(tap,ci)][co] += A[(tap,ci)][voxel] * B[voxel][co]
//       workgroup = 2x4x16 voxels per step, x halo tile and dy tile transposed in LDS
//       ([channel][voxel], odd half-stride -> conflict-free ds_read_b32); each workgroup walks a
//       strided list of voxel tiles with all 27 taps' accumulators in registers and flushes once with
//       float atomics.
// The data-gradient is the forward kernel on weights packed with flipped taps / swapped channels.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MAX_NT = 6;  // n-tiles (of 16 output channels) per workgroup

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// N (output channel) chunking shared by pack + forward
struct NChunk {
  int ntiles, nchunks, NT;
};
inline NChunk n_chunking(int Cout) {
  NChunk r;
  r.ntiles = cdiv(Cout, 16);
  r.nchunks = cdiv(r.ntiles, MAX_NT);
  r.NT = cdiv(r.ntiles, r.nchunks);
  return r;
}
inline int ck_for(int Cin) { return (Cin % 24 == 0) ? 24 : 8; }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// -------------------------------------------------------------------------------------------- pack
// packed[nc][cc][tap][cg][nt][lane][2]; lane=(kq=lane>>4, j=lane&15):
//   W_eff[tap][cc*CK + cg*8 + 2*kq + s][(nc*NT + nt)*16 + j]
__global__ void pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cin, int Cout, int mode, int CK,
                            int ncc, int NT, int nchunks, int64_t total) {
  const int CinE = mode ? Cout : Cin, CoutE = mode ? Cin : Cout;
  const int NCG = CK / 8;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx;
    const int s = (int)(r & 1);
    r >>= 1;
    const int lane = (int)(r & 63);
    r >>= 6;
    const int nt = (int)(r % NT);
    r /= NT;
    const int cg = (int)(r % NCG);
    r /= NCG;
    const int tap = (int)(r % 27);
    r /= 27;
    const int cc = (int)(r % ncc);
    const int nc = (int)(r / ncc);
    const int kq = lane >> 4, j = lane & 15;
    const int ci = cc * CK + cg * 8 + 2 * kq + s;
    const int co = (nc * NT + nt) * 16 + j;
    float v = 0.f;
    if (ci < CinE && co < CoutE) {
      if (mode == 0)
        v = w[((int64_t)tap * Cin + ci) * Cout + co];
      else
        v = w[((int64_t)(26 - tap) * Cin + co) * Cout + ci];
    }
    packed[idx] = v;
  }
}

// -------------------------------------------------------------------------------------------- forward
constexpr int FT0 = 4, FT1 = 4, FT2 = 16;              // output tile
constexpr int FH0 = 6, FH1 = 6, FH2 = 18;              // halo tile
constexpr int FHV = FH0 * FH1 * FH2;                   // 648 halo voxels

template <int CK, int NT>
__global__ __launch_bounds__(256, 2) void conv3d_fwd_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int D0, int D1, int D2, int Cin, int Cout, int ncc,
                                                            int tiles1, int tiles2, int act) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [FHV][CKP]
  constexpr int CKP = CK + 4;
  constexpr int NCG = CK / 8;
  constexpr int C4 = CK / 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int t = blockIdx.x;
  const int t2 = t % tiles2;
  t /= tiles2;
  const int t1 = t % tiles1;
  const int t0 = t / tiles1;
  const int z0 = t0 * FT0, y0 = t1 * FT1, x0 = t2 * FT2;
  const int nc = blockIdx.y;

  f32x4 acc[4][NT];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int li = lane & 15, kq = lane >> 4;
  int a_base[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) a_base[m] = ((wave * FH1 + m) * FH2 + li) * CKP + 2 * kq;

  const float* wl = wp + (size_t)nc * ncc * 27 * NCG * NT * 128 + lane * 2;
  const bool vec_ok = (Cin % 4) == 0;

  for (int cc = 0; cc < ncc; ++cc) {
    __syncthreads();
    // ---- stage the halo tile of this channel chunk (zero padding outside the volume / channel range)
    for (int f = tid; f < FHV * C4; f += 256) {
      const int vox = f / C4, c4 = f - vox * C4;
      const int hx = vox % FH2, hy = (vox / FH2) % FH1, hz = vox / (FH2 * FH1);
      const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gz >= 0 && gz < D0 && gy >= 0 && gy < D1 && gx >= 0 && gx < D2) {
        const int c = cc * CK + c4 * 4;
        const float* src = in + (((size_t)gz * D1 + gy) * D2 + gx) * Cin + c;
        if (vec_ok) {
          if (c < Cin) v = ld4(src);
        } else {
          if (c + 0 < Cin) v.x = src[0];
          if (c + 1 < Cin) v.y = src[1];
          if (c + 2 < Cin) v.z = src[2];
          if (c + 3 < Cin) v.w = src[3];
        }
      }
      *reinterpret_cast<float4*>(&lds[vox * CKP + c4 * 4]) = v;
    }
    __syncthreads();

    const float* wc = wl + (size_t)cc * 27 * NCG * NT * 128;
    float2 bcur[NCG][NT], bnext[NCG][NT];
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
      for (int n = 0; n < NT; ++n) bcur[g][n] = *reinterpret_cast<const float2*>(wc + (g * NT + n) * 128);

    for (int tap = 0; tap < 27; ++tap) {
      if (tap + 1 < 27) {
        const float* wn = wc + (size_t)(tap + 1) * NCG * NT * 128;
#pragma unroll
        for (int g = 0; g < NCG; ++g)
#pragma unroll
          for (int n = 0; n < NT; ++n) bnext[g][n] = *reinterpret_cast<const float2*>(wn + (g * NT + n) * 128);
      }
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
      const int toff = ((dz * FH1 + dy) * FH2 + dx) * CKP;
#pragma unroll
      for (int g = 0; g < NCG; ++g) {
        float2 a[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) a[m] = *reinterpret_cast<const float2*>(&lds[a_base[m] + toff + g * 8]);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int m = 0; m < 4; ++m)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].x, bcur[g][n].x, acc[m][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int m = 0; m < 4; ++m)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].y, bcur[g][n].y, acc[m][n], 0, 0, 0);
      }
      if (tap + 1 < 27) {
#pragma unroll
        for (int g = 0; g < NCG; ++g)
#pragma unroll
          for (int n = 0; n < NT; ++n) bcur[g][n] = bnext[g][n];
      }
    }
  }

  // ---- epilogue: D row = (lane>>4)*4 + reg -> x within the 16-voxel row, col = lane&15 -> output channel
  const int gz = z0 + wave;
  if (gz < D0) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int gy = y0 + m;
      if (gy >= D1) continue;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int co = (nc * NT + n) * 16 + li;
        if (co >= Cout) continue;
        const float bv = bias ? bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gx = x0 + kq * 4 + r;
          if (gx < D2) {
            float v = acc[m][n][r] + bv;
            if (act == 1) v = v > 0.f ? v : expm1f(v);
            out[(((size_t)gz * D1 + gy) * D2 + gx) * Cout + co] = v;
          }
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------- weight gradient
constexpr int WT0 = 2, WT1 = 4, WT2 = 16;  // voxel tile = 128 voxels = 32 k-steps
constexpr int WH0 = 4, WH1 = 6, WH2 = 18;
constexpr int WHV = WH0 * WH1 * WH2;        // 432
constexpr int WVPX = 434;                   // 434/2 = 217 odd -> rows of [ci][voxel] land on distinct even banks
constexpr int WTV = WT0 * WT1 * WT2;        // 128
constexpr int WVPD = 130;                   // 65 odd

template <int CK, int NT>
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_kernel(const float* __restrict__ in,
                                                              const float* __restrict__ dout, float* __restrict__ dw,
                                                              int D0, int D1, int D2, int Cin, int Cout, int tiles0,
                                                              int tiles1, int tiles2) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lx = lds;                 // [CK][WVPX]
  float* ld = lds + CK * WVPX;     // [NT*16][WVPD]
  constexpr int MR = 27 * CK;      // GEMM rows (tap, ci)
  constexpr int MTILES = (MR + 15) / 16;
  constexpr int MTW = (MTILES + 3) / 4;  // m-tiles per wave
  constexpr int C4 = CK / 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int cc = blockIdx.y;   // input-channel chunk
  const int nco = blockIdx.z;  // output-channel chunk
  const int co0 = nco * NT * 16;
  const bool vec_in = (Cin % 4) == 0, vec_out = (Cout % 4) == 0;

  f32x4 acc[MTW][NT];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-lane A base offsets: row r = (wave + 4*m)*16 + li -> (tap, cil)
  int a_base[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    int r = (wave + 4 * m) * 16 + li;
    if (r >= MR) r = MR - 1;
    const int tap = r / CK, cil = r - tap * CK;
    const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
    a_base[m] = cil * WVPX + (dz * WH1 + dy) * WH2 + dx;
  }
  const int b_base = li * WVPD + kq;

  const int ntiles = tiles0 * tiles1 * tiles2;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int t2 = t % tiles2, t1 = (t / tiles2) % tiles1, t0 = t / (tiles2 * tiles1);
    const int z0 = t0 * WT0, y0 = t1 * WT1, x0 = t2 * WT2;
    __syncthreads();
    // ---- stage x halo tile, transposed to [ci][voxel]
    for (int f = tid; f < WHV * C4; f += 256) {
      const int vox = f / C4, c4 = f - vox * C4;
      const int hx = vox % WH2, hy = (vox / WH2) % WH1, hz = vox / (WH2 * WH1);
      const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gz >= 0 && gz < D0 && gy >= 0 && gy < D1 && gx >= 0 && gx < D2) {
        const int c = cc * CK + c4 * 4;
        const float* src = in + (((size_t)gz * D1 + gy) * D2 + gx) * Cin + c;
        if (vec_in) {
          if (c < Cin) v = ld4(src);
        } else {
          if (c + 0 < Cin) v.x = src[0];
          if (c + 1 < Cin) v.y = src[1];
          if (c + 2 < Cin) v.z = src[2];
          if (c + 3 < Cin) v.w = src[3];
        }
      }
      float* d = lx + (c4 * 4) * WVPX + vox;
      d[0] = v.x;
      d[WVPX] = v.y;
      d[2 * WVPX] = v.z;
      d[3 * WVPX] = v.w;
    }
    // ---- stage dy tile, transposed to [co][voxel]
    for (int f = tid; f < WTV * NT * 4; f += 256) {
      const int vox = f / (NT * 4), c4 = f - vox * (NT * 4);
      const int vx = vox % WT2, vy = (vox / WT2) % WT1, vz = vox / (WT2 * WT1);
      const int gz = z0 + vz, gy = y0 + vy, gx = x0 + vx;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gz < D0 && gy < D1 && gx < D2) {
        const int c = co0 + c4 * 4;
        const float* src = dout + (((size_t)gz * D1 + gy) * D2 + gx) * Cout + c;
        if (vec_out) {
          if (c < Cout) v = ld4(src);
        } else {
          if (c + 0 < Cout) v.x = src[0];
          if (c + 1 < Cout) v.y = src[1];
          if (c + 2 < Cout) v.z = src[2];
          if (c + 3 < Cout) v.w = src[3];
        }
      }
      float* d = ld + (c4 * 4) * WVPD + vox;
      d[0] = v.x;
      d[WVPD] = v.y;
      d[2 * WVPD] = v.z;
      d[3 * WVPD] = v.w;
    }
    __syncthreads();
    // ---- 32 k-steps of 4 voxels
#pragma unroll 2
    for (int ks = 0; ks < WTV / 4; ++ks) {
      const int k = ks * 4 + kq;  // this lane's voxel within the tile
      const int vx = k & 15, vy = (k >> 4) & 3, vz = k >> 6;
      const int voff = (vz * WH1 + vy) * WH2 + vx;
      float b[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) b[n] = ld[b_base + n * 16 * WVPD + ks * 4];
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        const float a = lx[a_base[m] + voff];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[n], acc[m][n], 0, 0, 0);
      }
    }
  }

  // ---- flush: D row = (lane>>4)*4 + reg -> (tap, ci), col = lane&15 -> co
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = (wave + 4 * m) * 16 + kq * 4 + r;
      if (row >= MR) continue;
      const int tap = row / CK, cil = row - tap * CK;
      const int ci = cc * CK + cil;
      if (ci >= Cin) continue;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int co = co0 + n * 16 + li;
        if (co < Cout) atomicAdd(&dw[((size_t)tap * Cin + ci) * Cout + co], acc[m][n][r]);
      }
    }
  }
}

template <int CK, int NT>
int launch_fwd(const float* in, const float* wp, const float* bias, float* out, const int s[3], int Cin, int Cout,
               int ncc, int nchunks, int act, hipStream_t st) {
  const int tiles0 = cdiv(s[0], FT0), tiles1 = cdiv(s[1], FT1), tiles2 = cdiv(s[2], FT2);
  const size_t smem = (size_t)FHV * (CK + 4) * sizeof(float);
  static bool attr_done = false;
  auto kern = conv3d_fwd_kernel<CK, NT>;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles0 * tiles1 * tiles2, nchunks), dim3(256), smem, st, in, wp, bias, out, s[0], s[1],
                     s[2], Cin, Cout, ncc, tiles1, tiles2, act);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

template <int CK>
int dispatch_fwd(int NT, const float* in, const float* wp, const float* bias, float* out, const int s[3], int Cin,
                 int Cout, int ncc, int nchunks, int act, hipStream_t st) {
  switch (NT) {
    case 1: return launch_fwd<CK, 1>(in, wp, bias, out, s, Cin, Cout, ncc, nchunks, act, st);
    case 2: return launch_fwd<CK, 2>(in, wp, bias, out, s, Cin, Cout, ncc, nchunks, act, st);
    case 3: return launch_fwd<CK, 3>(in, wp, bias, out, s, Cin, Cout, ncc, nchunks, act, st);
    case 4: return launch_fwd<CK, 4>(in, wp, bias, out, s, Cin, Cout, ncc, nchunks, act, st);
    case 5: return launch_fwd<CK, 5>(in, wp, bias, out, s, Cin, Cout, ncc, nchunks, act, st);
    case 6: return launch_fwd<CK, 6>(in, wp, bias, out, s, Cin, Cout, ncc, nchunks, act, st);
  }
  return SYNTHSR_EINVAL;
}

template <int CK, int NT>
int launch_wgrad(const float* in, const float* dout, float* dw, const int s[3], int Cin, int Cout, hipStream_t st) {
  const int tiles0 = cdiv(s[0], WT0), tiles1 = cdiv(s[1], WT1), tiles2 = cdiv(s[2], WT2);
  const int ntiles = tiles0 * tiles1 * tiles2;
  const int ncc = cdiv(Cin, CK), nco = cdiv(Cout, NT * 16);
  int gx = 2048 / (ncc * nco);
  if (gx < 1) gx = 1;
  if (gx > ntiles) gx = ntiles;
  const size_t smem = ((size_t)CK * WVPX + (size_t)NT * 16 * WVPD) * sizeof(float);
  static bool attr_done = false;
  auto kern = conv3d_wgrad_kernel<CK, NT>;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(gx, ncc, nco), dim3(256), smem, st, in, dout, dw, s[0], s[1], s[2], Cin, Cout, tiles0,
                     tiles1, tiles2);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

}  // namespace

extern "C" {

int64_t synthsr_conv3d_pack(const float* w, float* packed, int Cin, int Cout, int mode, synthsr_stream_t stream) {
  if (Cin < 1 || Cout < 1 || (mode != 0 && mode != 1)) return SYNTHSR_EINVAL;
  const int CinE = mode ? Cout : Cin, CoutE = mode ? Cin : Cout;
  const int CK = ck_for(CinE);
  const int ncc = cdiv(CinE, CK);
  const NChunk nch = n_chunking(CoutE);
  const int64_t total = (int64_t)nch.nchunks * ncc * 27 * (CK / 8) * nch.NT * 128;
  if (!packed) return total;
  if (!w) return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(pack_kernel, dim3(syn_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w, packed, Cin, Cout,
                     mode, CK, ncc, nch.NT, nch.nchunks, total);
  if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
  return total;
}

int synthsr_conv3d_fwd(const float* in, const float* wpacked, const float* bias, float* out, const int shape[3], int Cin,
                       int Cout, int act, synthsr_stream_t stream) {
  if (!in || !wpacked || !out || !shape || Cin < 1 || Cout < 1 || shape[0] < 1 || shape[1] < 1 || shape[2] < 1 ||
      (act != 0 && act != 1))
    return SYNTHSR_EINVAL;
  const int CK = ck_for(Cin);
  const int ncc = cdiv(Cin, CK);
  const NChunk nch = n_chunking(Cout);
  if (CK == 24)
    return dispatch_fwd<24>(nch.NT, in, wpacked, bias, out, shape, Cin, Cout, ncc, nch.nchunks, act, (hipStream_t)stream);
  return dispatch_fwd<8>(nch.NT, in, wpacked, bias, out, shape, Cin, Cout, ncc, nch.nchunks, act, (hipStream_t)stream);
}

int synthsr_conv3d_wgrad(const float* in, const float* dout, float* dw, const int shape[3], int Cin, int Cout,
                         synthsr_stream_t stream) {
  if (!in || !dout || !dw || !shape || Cin < 1 || Cout < 1 || shape[0] < 1 || shape[1] < 1 || shape[2] < 1)
    return SYNTHSR_EINVAL;
  const int CK = ck_for(Cin);
  // output-channel chunks of <= 48 (3 n-tiles) keep 27 taps x 24 ci of accumulators in registers
  const int nt_all = cdiv(Cout, 16);
  const int nco = cdiv(nt_all, 3);
  const int NT = cdiv(nt_all, nco);
  hipStream_t st = (hipStream_t)stream;
  if (CK == 24) {
    if (NT == 1) return launch_wgrad<24, 1>(in, dout, dw, shape, Cin, Cout, st);
    if (NT == 2) return launch_wgrad<24, 2>(in, dout, dw, shape, Cin, Cout, st);
    return launch_wgrad<24, 3>(in, dout, dw, shape, Cin, Cout, st);
  }
  if (NT == 1) return launch_wgrad<8, 1>(in, dout, dw, shape, Cin, Cout, st);
  if (NT == 2) return launch_wgrad<8, 2>(in, dout, dw, shape, Cin, Cout, st);
  return launch_wgrad<8, 3>(in, dout, dw, shape, Cin, Cout, st);
}

}  // extern "C"
