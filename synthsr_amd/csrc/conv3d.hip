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
#include <algorithm>
#include <cstddef>

SYN_DET_SETTER(conv3d)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MAX_NT = 6;  // n-tiles (of 16 output channels) per workgroup

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

inline int ck_for(int Cin) { return (Cin % 24 == 0) ? 24 : 8; }
// forward / data-gradient kernels also take 32-channel chunks (the critic's 32 / 64 / 128 / 256 channels: four 8-channel
// chunks re-staged the halo tile four times per tile and ran at 6 % of the MFMA peak)
inline int ck_for_fwd(int Cin) { return (Cin % 24 == 0) ? 24 : ((Cin % 32 == 0) ? 32 : 8); }

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {  // compile-time loop: f(std::integral_constant<int, I>) for I in [I, N)
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// ELU(alpha = 1) for the fused epilogues.  Vector-ALU instructions next to MFMAs are not free on gfx950 (see the
// persistent kernel), so instead of libm's expm1f (~22 instructions) the negative branch is 2^(v log2 e) - 1 through
// v_exp_f32, switched to a degree-5 Taylor polynomial on (-1/8, 0] where the subtraction would cancel.
// |error| < 2e-7 absolute and < 2e-6 relative to expm1 (tests/test_unet_gpu.py::test_elu_epilogue_accuracy).
__device__ __forceinline__ float elu_f(float v) {
  const float e = __builtin_amdgcn_exp2f(v * 1.44269504088896341f) - 1.f;
  const float p = v * fmaf(v, fmaf(v, fmaf(v, fmaf(v, 1.f / 120.f, 1.f / 24.f), 1.f / 6.f), 0.5f), 1.f);
  const float n = v > -0.125f ? p : e;
  return v > 0.f ? v : n;
}
// derivative of ELU expressed through its output y: 1 for y > 0, y + 1 (= e^x) otherwise
__device__ __forceinline__ float elu_dy(float y) { return y > 0.f ? 1.f : y + 1.f; }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// -------------------------------------------------------------------------------------------- pack
// packed[nc][cc][tap][cg][nt][lane][2]; lane=(kq=lane>>4, j=lane&15):
//   W_eff[tap][cc*CK + cg*8 + 2*kq + s][(nc*NT + nt)*16 + j]
// taps of the original 3-tap axis that fold onto low-res slot s under output parity p (see ConvExt)
__device__ __forceinline__ int up_axis_taps(int p, int s, int t[2]) {
  if (p == 0) {
    if (s == 0) { t[0] = 0; return 1; }
    if (s == 1) { t[0] = 1; t[1] = 2; return 2; }
    return 0;
  }
  if (s == 1) { t[0] = 0; t[1] = 1; return 2; }
  if (s == 2) { t[0] = 2; return 1; }
  return 0;
}

// A stride-2 'same' conv of an even-sized volume (TensorFlow pads (0, 1): output o reads inputs 2o, 2o+1, 2o+2) is also a
// sum of 8 parity convs on the low-res grid: with x_p[i] = x[2 i + p], tap 0 -> (p 0, offset 0), tap 1 -> (p 1, offset 0),
// tap 2 -> (p 0, offset +1).  In 27-slot form (slot s <-> offset s - 1) the original tap behind slot s of parity p:
__device__ __forceinline__ int stride_axis_taps(int p, int s, int t[2]) {
  if (p == 0 && s == 1) { t[0] = 0; return 1; }
  if (p == 0 && s == 2) { t[0] = 2; return 1; }
  if (p == 1 && s == 1) { t[0] = 1; return 1; }
  return 0;
}
// These slots lie inside the 2x2x2 windows of the folded decoder conv's data gradient (mode 2: slots {1-p, 2-p}) and, after
// the mode-1 flip, of its forward pass ({p, p+1}), so the strided layer runs on those kernels unchanged: forward through
// synthsr_conv3d_up_dgrad, data gradient through synthsr_conv3d_up_fwd, with weight sets packed under parity codes 8..15.

// w: Keras kernel [27][Cin_total][Cout]; the layer (or layer part) uses input channels [ci_off, ci_off+Cin).
// parity < 0: plain weights.  parity 0..7: combined weights of the nearest-upsample folding in 27-slot form.
// value of the effective weight W_eff[tap][cie][coe] of a (possibly transposed / parity-folded) layer part
__device__ __forceinline__ float weight_value(const float* __restrict__ w, int tap, int cie, int coe, int Cin_total,
                                              int ci_off, int Cin, int Cout, int mode, int parity) {
  const int CinE = mode ? Cout : Cin, CoutE = mode ? Cin : Cout;
  if (cie >= CinE || coe >= CoutE) return 0.f;
  const int slot = mode ? 26 - tap : tap;                            // tap slot in forward orientation
  const int ci = (mode ? coe : cie) + ci_off, co = mode ? cie : coe;  // layer channel indices
  if (parity < 0) return w[((int64_t)slot * Cin_total + ci) * Cout + co];
  int tz[2], ty[2], tx[2];
  const bool strided = parity >= 8;  // 8 + p: the parity sets of a stride-2 conv
  const int nz = strided ? stride_axis_taps((parity >> 2) & 1, slot / 9, tz) : up_axis_taps((parity >> 2) & 1, slot / 9, tz);
  const int ny = strided ? stride_axis_taps((parity >> 1) & 1, (slot / 3) % 3, ty)
                         : up_axis_taps((parity >> 1) & 1, (slot / 3) % 3, ty);
  const int nx = strided ? stride_axis_taps(parity & 1, slot % 3, tx) : up_axis_taps(parity & 1, slot % 3, tx);
  float v = 0.f;
  for (int a = 0; a < nz; ++a)
    for (int b = 0; b < ny; ++b)
      for (int c = 0; c < nx; ++c) v += w[((int64_t)((tz[a] * 3 + ty[b]) * 3 + tx[c]) * Cin_total + ci) * Cout + co];
  return v;
}

// ---- split layouts (conv_split.hip): bf16 A fragments of the three pieces of every weight, two bf16 per float slot --
// [piece 3][co-chunk][cc][step][mt][lane 64][8]; lane = (m = lane & 15 -> output channel, g = lane >> 4 -> K slot 4 step + g),
// value j -> input channel cc*8 + j.  NT = -100 - MT: plain conv, 7 K steps, slot -> tap by syn_split_tap (one spare slot:
// zero).  NT = -200 - MT / -300 - MT: ONE parity set of a folded decoder conv, 2 K steps whose slots are the 8 taps (a, b, c)
// of the parity's 2x2x2 window (syn_split_tap8); the kernel reads tap (a, b, c) at halo offset (1 - p) + a per axis (UPM 2:
// folded data gradient, or a stride-2 forward, parity code 8 + p) or a + p (the forward kernel conv3d_split_upfwd_kernel) --
// that offset IS the 27-slot index weight_value folds the original taps onto, whatever the orientation (`mode`) of the set.
// `nchunks` (the job's mfma_count field) = number of co-chunks.
__host__ __device__ inline int64_t split_plane_floats(int NT, int ncc, int nchunks) {  // floats of one piece plane
  const int MT = NT <= -300 ? -300 - NT : (NT <= -200 ? -200 - NT : -100 - NT);
  return (int64_t)nchunks * ncc * (NT <= -200 ? 2 : 7) * MT * 64 * 4;
}
// the three pieces (packed bf16 pairs) of float slot i of a piece plane
__device__ __forceinline__ void split_pack_pair(const float* __restrict__ w, int64_t i, int Cin_total, int ci_off, int Cin, int Cout,
                                                int mode, int ncc, int NT, int parity, int nchunks, uint32_t pc[3]) {
  const bool up = NT <= -200, fwdwin = NT <= -300;
  const int MT = fwdwin ? -300 - NT : (up ? -200 - NT : -100 - NT), nstep = up ? 2 : 7;
  uint32_t r = (uint32_t)i * 2u;
  const int j = (int)(r & 7);
  r >>= 3;
  const int lane = (int)(r & 63);
  r >>= 6;
  const int mt = (int)(r % MT);
  r /= MT;
  const int step = (int)(r % nstep);
  r /= nstep;
  const int cc = (int)(r % ncc);
  const int chunk = (int)(r / ncc);
  (void)nchunks;
  const int coe = (chunk * MT + mt) * 16 + (lane & 15);
  int tap;
  if (up) {
    const int t8 = syn_split_tap8(4 * step + (lane >> 4));
    const int pz = (parity >> 2) & 1, py = (parity >> 1) & 1, px = parity & 1;
    const int hz = (fwdwin ? pz : 1 - pz) + (t8 >> 2), hy = (fwdwin ? py : 1 - py) + ((t8 >> 1) & 1),
              hx = (fwdwin ? px : 1 - px) + (t8 & 1);
    tap = (hz * 3 + hy) * 3 + hx;
  } else {
    tap = syn_split_tap(4 * step + (lane >> 4));
  }
  float v0 = 0.f, v1 = 0.f;
  if (tap >= 0) {
    v0 = weight_value(w, tap, cc * 8 + j, coe, Cin_total, ci_off, Cin, Cout, mode, parity);
    v1 = weight_value(w, tap, cc * 8 + j + 1, coe, Cin_total, ci_off, Cin, Cout, mode, parity);
  }
  syn_split3(v0, v1, pc[0], pc[1], pc[2]);
}

// Stacked layout of the plain Cout = 24 split convs (NT = -400; conv_split.hip: conv3d_split_fwd2_kernel<2, ., ., true>):
// [cc][step 7][tile 5][lane 64][8 bf16]; lane = (m = lane & 15 -> row of the tile, g = lane >> 4 -> K slot 4 step + g), value j ->
// input channel cc*8 + j.  Rows: T0 = piece 0 of channels 0..15, T1 = piece 1, T2 = piece 2 of the same channels; T3 = piece 0 of
// channels 16..23 (rows 0..7) | piece 1 of channels 16..23 (rows 8..15); T4 = piece 2 of channels 16..23 (rows 0..7) | zeros.
__device__ __forceinline__ float stacked_pack_value(const float* __restrict__ w, int64_t idx, int Cin_total, int ci_off, int Cin,
                                                    int Cout, int mode) {
  uint32_t r = (uint32_t)idx * 2u;  // bf16 index
  const int j = (int)(r & 7);
  r >>= 3;
  const int lane = (int)(r & 63);
  r >>= 6;
  const int tile = (int)(r % 5);
  r /= 5;
  const int step = (int)(r % 7);
  const int cc = (int)(r / 7);
  const int m = lane & 15;
  const int tap = syn_split_tap(4 * step + (lane >> 4));
  int piece, co;
  if (tile < 3) { piece = tile; co = m; }
  else if (tile == 3) { piece = m < 8 ? 0 : 1; co = 16 + (m & 7); }
  else { piece = m < 8 ? 2 : -1; co = 16 + (m & 7); }
  if (tap < 0 || piece < 0) return 0.f;
  const float v0 = weight_value(w, tap, cc * 8 + j, co, Cin_total, ci_off, Cin, Cout, mode, -1);
  const float v1 = weight_value(w, tap, cc * 8 + j + 1, co, Cin_total, ci_off, Cin, Cout, mode, -1);
  uint32_t pc[3];
  syn_split3(v0, v1, pc[0], pc[1], pc[2]);
  return __uint_as_float(pc[piece]);
}

// packed layout of one weight set: MFMA section [nc][cc][tap][cg][nt][lane][2] (B fragments), followed — when the layer
// keeps NV output channels on the vector ALUs — by the VALU section [cc][tap][ci (CK)][NV] (wave-uniform scalar loads)
__device__ __forceinline__ float pack_value(const float* __restrict__ w, int64_t idx, int Cin_total, int ci_off, int Cin,
                                            int Cout, int mode, int CK, int ncc, int NT, int parity, int NV,
                                            int64_t mfma_count) {
  if (NT == -400) return stacked_pack_value(w, idx, Cin_total, ci_off, Cin, Cout, mode);
  if (NT <= -100) {  // split layouts: one piece of split_pack_pair
    const int64_t n3 = split_plane_floats(NT, ncc, (int)mfma_count);
    uint32_t pc[3];
    split_pack_pair(w, idx % n3, Cin_total, ci_off, Cin, Cout, mode, ncc, NT, parity, (int)mfma_count, pc);
    return __uint_as_float(pc[idx / n3]);
  }
  if (NT < 0) {
    // first-layer layout (conv3d_fwd_c2_kernel, Cin = -NT <= 2, Cout = 24): [r][lane 64], G = r*16 + (lane >> 2) =
    // k*6 + g with k = tap*Cin + ci; output channels 4g + (lane & 3); zero beyond k = 27*Cin
    const int cin = -NT;
    const int lane = (int)(idx & 63);
    const int G = (int)(idx >> 6) * 16 + (lane >> 2);
    const int k = G / 6, g = G % 6;
    if (k >= 27 * cin) return 0.f;
    return weight_value(w, k / cin, k % cin, g * 4 + (lane & 3), Cin_total, ci_off, Cin, Cout, mode, parity);
  }
  if (NT == 0) {
    // 4x4x1-MFMA layout of the Cout = 24 layers (conv3d_fwd_p4_kernel): [cc][tap][qp 3][r 3][lane 64].  Register r of
    // channel-octet qp holds 16 four-channel groups, G = r*16 + (lane >> 2) = h*24 + kk*6 + g  ->  input channel
    // cc*24 + qp*8 + h*4 + kk, output channels 4g + (lane & 3); the kernel selects a group with the MFMA's ABID.
    const int lane = (int)(idx & 63);
    uint32_t r = (uint32_t)(idx >> 6);  // 32-bit index arithmetic: a weight set has < 2^31 values (checked by the launcher)
    const int rr = (int)(r % 3);
    r /= 3;
    const int qp = (int)(r % 3);
    r /= 3;
    const int tap = (int)(r % 27);
    const int cc = (int)(r / 27);
    const int G = rr * 16 + (lane >> 2);
    const int cie = cc * 24 + qp * 8 + (G / 24) * 4 + (G % 24) / 6;
    const int coe = (G % 6) * 4 + (lane & 3);
    return weight_value(w, tap, cie, coe, Cin_total, ci_off, Cin, Cout, mode, parity);
  }
  if (idx >= mfma_count) {
    uint32_t r = (uint32_t)(idx - mfma_count);
    const int v = (int)(r % NV);
    r /= NV;
    const int cil = (int)(r % CK);
    r /= CK;
    const int tap = (int)(r % 27);
    const int cc = (int)(r / 27);
    const int coutE = mode ? Cin : Cout;
    return weight_value(w, tap, cc * CK + cil, coutE - NV + v, Cin_total, ci_off, Cin, Cout, mode, parity);
  }
  const int NCG = CK / 8;
  uint32_t r = (uint32_t)idx;
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
  const int cie = cc * CK + cg * 8 + 2 * kq + s;
  const int coe = (nc * NT + nt) * 16 + j;
  const int coutE = mode ? Cin : Cout;
  if (coe >= coutE - NV) return 0.f;  // channels owned by the VALU section
  return weight_value(w, tap, cie, coe, Cin_total, ci_off, Cin, Cout, mode, parity);
}

__global__ void pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cin_total, int ci_off, int Cin,
                            int Cout, int mode, int CK, int ncc, int NT, int nchunks, int parity, int NV,
                            int64_t mfma_count, int64_t total) {
  if (NT <= -100 && NT != -400) {  // split layouts: a thread gathers a weight pair once and writes its three pieces
    const int64_t n3 = total / 3;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n3; i += (int64_t)gridDim.x * blockDim.x) {
      uint32_t pc[3];
      split_pack_pair(w, i, Cin_total, ci_off, Cin, Cout, mode, ncc, NT, parity, (int)mfma_count, pc);
      packed[i] = __uint_as_float(pc[0]);
      packed[i + n3] = __uint_as_float(pc[1]);
      packed[i + 2 * n3] = __uint_as_float(pc[2]);
    }
    return;
  }
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x)
    packed[idx] = pack_value(w, idx, Cin_total, ci_off, Cin, Cout, mode, CK, ncc, NT, parity, NV, mfma_count);
}

// one launch for every layer of the network: jobs[j] = {w_off, dst_off, count, cin_total, ci_off, cin, cout, mode, ck,
// ncc, nt, parity, nv, mfma_count} (int64 each), blockIdx.y = job
constexpr int PACK_JOB_FIELDS = 14;
__global__ void pack_all_kernel(const float* __restrict__ params, float* __restrict__ packed,
                                const int64_t* __restrict__ jobs) {
  const int64_t* jb = jobs + (int64_t)blockIdx.y * PACK_JOB_FIELDS;
  const float* w = params + jb[0];
  float* dst = packed + jb[1];
  const int64_t count = jb[2];
  const int cin_total = (int)jb[3], ci_off = (int)jb[4], cin = (int)jb[5], cout = (int)jb[6], mode = (int)jb[7],
            ck = (int)jb[8], ncc = (int)jb[9], nt = (int)jb[10], parity = (int)jb[11], nv = (int)jb[12];
  const int64_t mfma_count = jb[13];
  if (nt <= -100 && nt != -400) {  // split layouts: a thread gathers a weight pair once and writes its three pieces
    const int64_t n3 = count / 3;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n3; i += (int64_t)gridDim.x * blockDim.x) {
      uint32_t pc[3];
      split_pack_pair(w, i, cin_total, ci_off, cin, cout, mode, ncc, nt, parity, (int)mfma_count, pc);
      dst[i] = __uint_as_float(pc[0]);
      dst[i + n3] = __uint_as_float(pc[1]);
      dst[i + 2 * n3] = __uint_as_float(pc[2]);
    }
    return;
  }
  if (nt > 0 && parity >= 0 && parity < 8 && nv == 0) {
    // one parity set of a folded decoder conv in the 27-slot MFMA layout [nc][cc][slot 27][cg][nt][lane][2]: only the 8 slots of
    // the parity's 2x2x2 window (forward orientation: coordinates p .. p + 1 per axis, up_axis_taps) are ever non-zero, the
    // other 19 stay at the zeros the buffer was created with -- 143 of the 259 MB this kernel used to write per step (the
    // 10^3 384 -> 192 sets alone: 2 x 64 MB) were those zeros
    const int64_t inner = (int64_t)(ck / 8) * nt * 128, count8 = count / 27 * 8;
    const int pz = (parity >> 2) & 1, py = (parity >> 1) & 1, px = parity & 1;
    for (int64_t i8 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i8 < count8; i8 += (int64_t)gridDim.x * blockDim.x) {
      const int64_t q = i8 / inner, rem = i8 - q * inner;
      const int t8 = (int)(q & 7);
      const int slot = ((pz + (t8 >> 2)) * 3 + py + ((t8 >> 1) & 1)) * 3 + px + (t8 & 1);
      const int64_t idx = ((q >> 3) * 27 + (mode ? 26 - slot : slot)) * inner + rem;
      dst[idx] = pack_value(w, idx, cin_total, ci_off, cin, cout, mode, ck, ncc, nt, parity, nv, mfma_count);
    }
    return;
  }
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < count;
       idx += (int64_t)gridDim.x * blockDim.x)
    dst[idx] = pack_value(w, idx, cin_total, ci_off, cin, cout, mode, ck, ncc, nt, parity, nv, mfma_count);
}

// dW of the up-sampled input channels from the 8 per-parity 27-slot gradients: every original tap t belongs to exactly
// one slot per parity.  dw[27][Cin_total][Cout] (+= at channels [ci_off, ci_off+Cl)), dwc[8][27][Cl][Cout].
// Thread = (slot triple s in {0,1,2}^3, ci, co): the taps of s per axis are {0} | {1} | {2} for s = 0 | 1 | 2 under parity
// 0 | either | 1 ... i.e. tap t reads slot s(p, t); equivalently every (parity, slot) value is read by the taps that fold onto it.
// Round 6: the partials are CONSUMED -- a second launch (same grid) writes zeros over the 8 x 8 slots the weight-gradient kernels
// can have written, so a caller that zeroed dwc once never has to again (85 MB of memsets per training step at configs[1]).
__global__ void up_unpack_kernel(const float* __restrict__ dwc, float* __restrict__ dw, int Cin_total, int ci_off, int Cl,
                                 int Cout, int64_t total) {
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(idx % Cout);
    const int ci = (int)((idx / Cout) % Cl);
    const int t = (int)(idx / ((int64_t)Cout * Cl));
    const int tz = t / 9, ty = (t / 3) % 3, tx = t % 3;
    float acc = 0.f;
    for (int p = 0; p < 8; ++p) {
      const int pz = (p >> 2) & 1, py = (p >> 1) & 1, px = p & 1;
      const int sz = pz ? (tz == 2 ? 2 : 1) : (tz == 0 ? 0 : 1);
      const int sy = py ? (ty == 2 ? 2 : 1) : (ty == 0 ? 0 : 1);
      const int sx = px ? (tx == 2 ? 2 : 1) : (tx == 0 ? 0 : 1);
      acc += dwc[(((int64_t)p * 27 + (sz * 3 + sy) * 3 + sx) * Cl + ci) * Cout + co];
    }
    dw[((int64_t)t * Cin_total + ci_off + ci) * Cout + co] += acc;
  }
}
// zeros over the slots p + {0, 1}^3 of every parity p (the only ones a folded weight-gradient kernel writes): 64 Cl Cout floats
__global__ void up_clear_kernel(float* __restrict__ dwc, int Cl, int Cout, int64_t total64) {
  const int64_t inner = (int64_t)Cl * Cout;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total64; idx += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(idx / inner);  // (parity, tap of its window)
    const int64_t rem = idx - (int64_t)q * inner;
    const int p = q >> 3, t8 = q & 7;
    const int slot = ((((p >> 2) & 1) + (t8 >> 2)) * 3 + ((p >> 1) & 1) + ((t8 >> 1) & 1)) * 3 + (p & 1) + (t8 & 1);
    dwc[((int64_t)p * 27 + slot) * inner + rem] = 0.f;
  }
}

// Epilogue of one wave: D fragments (row = (lane>>4)*4 + reg -> x, col = lane&15 -> channel) are transposed through a
// wave-private LDS slab so that each global store instruction writes 64 x 16 B of CONSECUTIVE addresses (a (z,y)
// row of 16 voxels x Cout channels is contiguous in NDHWC); direct fragment stores would emit 64-byte pieces.
// slab: 16 x (NT*16) floats.  Requires Cout % 4 == 0 (else the scalar path below).
template <int NT, int MT>
__device__ __forceinline__ void store_tile_rows(f32x4 (&acc)[MT][NT], float* slab, float* __restrict__ out,
                                                const float* __restrict__ bias, const float* addend, int act, int nc,
                                                int gz, int y0, int x0, int D1, int D2, int Cout, int lane) {
  const int li = lane & 15, kq = lane >> 4;
  constexpr int NW = NT * 16;
  const int c_lo = nc * NW;
  const int wc = min(Cout - c_lo, NW);  // channels of this chunk
  const int wc4 = wc >> 2;
  const int nx = min(16, D2 - x0);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int gy = y0 + m;
    if (gy >= D1) continue;  // wave-uniform
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int co = c_lo + n * 16 + li;
      const float bv = (bias && co < Cout) ? bias[co] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(kq * 4 + r) * NW + n * 16 + li] = acc[m][n][r] + bv;
    }
    __builtin_amdgcn_wave_barrier();
    const size_t rowoff = (((size_t)gz * D1 + gy) * D2 + x0) * Cout + c_lo;
    for (int idx = lane; idx < nx * wc4; idx += 64) {
      const int x = idx / wc4, c4 = idx - x * wc4;
      float4 v = *reinterpret_cast<const float4*>(&slab[x * NW + c4 * 4]);
      const size_t o = rowoff + (size_t)x * Cout + c4 * 4;
      if (act == 2) {  // data-gradient fused with the ELU backward of the producing layer: addend = its output y
        const float4 a = *reinterpret_cast<const float4*>(addend + o);
        v.x *= elu_dy(a.x);
        v.y *= elu_dy(a.y);
        v.z *= elu_dy(a.z);
        v.w *= elu_dy(a.w);
      } else if (addend) {  // wave-uniform; may alias out (same element read and written by this lane)
        const float4 a = *reinterpret_cast<const float4*>(addend + o);
        v.x += a.x;
        v.y += a.y;
        v.z += a.z;
        v.w += a.w;
      }
      if (act == 1) {
        v.x = elu_f(v.x);
        v.y = elu_f(v.y);
        v.z = elu_f(v.z);
        v.w = elu_f(v.w);
      }
      *reinterpret_cast<float4*>(out + o) = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// output tile = 4 (z, one per wave) x MT (y, m-tiles per wave) x 16 (x, MFMA rows); MT = 4 for the large levels,
// MT = 2 doubles the number of workgroups for the small (deep) levels.  KSPLIT: the input-channel chunks are
// split over gridDim.z workgroups that accumulate into a zero-initialised output with float atomics; bias and
// activation are then applied by bias_act_kernel.
constexpr int FT0 = 4, FT2 = 16;
constexpr int FH0 = 6, FH2 = 18;

// Nearest-upsample folding.  A 3x3x3 conv applied to UpSampling3D(2)(x) equals, for each output parity
// p = (pz,py,px), a 2x2x2 conv on x itself whose weights are sums of the original taps that read the same low-res
// voxel: per axis  p=0: slots {0 <- tap 0, 1 <- taps 1+2},  p=1: slots {1 <- taps 0+1, 2 <- tap 2}  (slot s reads
// low-res offset s-1).  3.4x fewer FLOPs on the up-sampled 2/3 of every decoder conv.  The parity convs run through
// the same kernels with a tap mask and strided input/output views.
struct ConvExt {
  int mode;            // 0 plain; 1 up-forward (parity = blockIdx.z, strided OUTPUT); 2 up-data-gradient (loop over
                       // the 8 parities, strided INPUT, one accumulated output)
  const float* addend; // mode 1: added before bias/activation, indexed like the output
  int64_t wstride;     // packed-weight stride between parities
  int64_t valu_off;    // offset of the VALU weight section inside one packed weight set
};

__host__ __device__ inline uint32_t up_tapmask(int p, bool flipped) {
  uint32_t m = 0;
  for (int sz = 0; sz < 3; ++sz)
    for (int sy = 0; sy < 3; ++sy)
      for (int sx = 0; sx < 3; ++sx) {
        const int pz = (p >> 2) & 1, py = (p >> 1) & 1, px = p & 1;
        const bool ok = (pz ? sz >= 1 : sz <= 1) && (py ? sy >= 1 : sy <= 1) && (px ? sx >= 1 : sx <= 1);
        if (ok) {
          const int t = flipped ? (((2 - sz) * 3 + (2 - sy)) * 3 + (2 - sx)) : ((sz * 3 + sy) * 3 + sx);
          m |= 1u << t;
        }
      }
  return m;
}

template <int CK, int NT, int MT, bool KSPLIT, int NV>
__global__ __launch_bounds__(256, 2) void conv3d_fwd_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int D0, int D1, int D2, int Cin, int Cout, int ncc,
                                                            int tiles1, int tiles2, int act, ConvExt ext) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [FHV][CKP]
  constexpr int FT1 = MT, FH1 = MT + 2, FHV = FH0 * FH1 * FH2;
  constexpr int CKP = CK + 4;
  constexpr int NCG = CK / 8;
  constexpr int C4 = CK / 4;
  const int dbg = act >> 8;  // diagnostic mask (synthsr_conv3d_set_option 1): 1 no B loads, 2 no A reads, 4 no staging, 8 no stores
  act &= 0xff;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware tile order: workgroup b runs on XCD b%8 (observed dispatch order, speed only); give every XCD a
  // contiguous range of tiles so that neighbouring tiles (shared halos) hit the same L2.  Bijective for any grid.
  int t;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, j = b >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int t2 = t % tiles2;
  t /= tiles2;
  const int t1 = t % tiles1;
  const int t0 = t / tiles1;
  const int z0 = t0 * FT0, y0 = t1 * FT1, x0 = t2 * FT2;
  const int nc = blockIdx.y;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // vector-ALU share (NV > 0): lane l owns voxel (z = wave, y = l>>4, x = l&15) of the tile and accumulates the last
  // NV output channels with v_fma, weights from SGPRs (VALU section of the packed buffer), in the shadow of the MFMAs
  static_assert(NV == 0 || (MT == 4 && !KSPLIT && CK == 24), "VALU share needs the 4x4x16 tile");
  float accv[NV > 0 ? NV : 1];
#pragma unroll
  for (int v = 0; v < (NV > 0 ? NV : 1); ++v) accv[v] = 0.f;
  const int vbase = ((wave * FH1 + (lane >> 4)) * FH2 + (lane & 15)) * CKP;

  const int li = lane & 15, kq = lane >> 4;
  int a_base[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) a_base[m] = ((wave * FH1 + m) * FH2 + li) * CKP + 2 * kq;

  const float* wl = wp + (size_t)nc * ncc * 27 * NCG * NT * 128 + lane * 2;

  // chunk range of this workgroup (all chunks unless KSPLIT)
  const int cpz = KSPLIT ? (ncc + (int)gridDim.z - 1) / (int)gridDim.z : ncc;
  const int cc_lo = KSPLIT ? (int)blockIdx.z * cpz : 0;
  const int cc_hi = min(ncc, cc_lo + cpz);
  const int npar = (ext.mode == 2) ? 8 : 1;          // parities accumulated inside this workgroup
  const int opar = (ext.mode == 1) ? (int)blockIdx.z : 0;  // output parity of this workgroup
  for (int it = 0; it < npar * (cc_hi - cc_lo); ++it) {
    const int ipar = it / (cc_hi - cc_lo);
    const int cc = cc_lo + it - ipar * (cc_hi - cc_lo);
    const int par = (ext.mode == 1) ? opar : ipar;
    const uint32_t tapmask = ext.mode == 0 ? 0x7FFFFFFu : up_tapmask(par, ext.mode == 2);
    // input view: conv-grid voxel g -> tensor voxel g*is + io (mode 2 reads one parity sub-lattice of a 2x tensor)
    const int is = (ext.mode == 2) ? 2 : 1;
    const int io0 = (ext.mode == 2) ? (par >> 2) & 1 : 0, io1 = (ext.mode == 2) ? (par >> 1) & 1 : 0,
              io2 = (ext.mode == 2) ? par & 1 : 0;
    __syncthreads();
    // ---- stage the halo tile of this channel chunk (zero padding outside the volume / channel range).
    // Branch-free: out-of-range elements load from a clamped (valid) address and are zeroed by a select, so all
    // global loads are issued back to back before the first LDS store and their latencies overlap.
    if (!(dbg & 4)) {
      constexpr int NIT = (FHV * C4 + 255) / 256;
      constexpr int SB = (NT >= 5 && CK == 24) ? (NIT + 1) / 2 : NIT;  // batch size (register budget)
#pragma unroll
      for (int k0 = 0; k0 < NIT; k0 += SB) {
        float4 stg[SB];
#pragma unroll
        for (int kk = 0; kk < SB; ++kk) {
          const int f = tid + (k0 + kk) * 256;
          const int vox = f / C4, c4 = f - vox * C4;
          const int hx = vox % FH2, hy = (vox / FH2) % FH1, hz = vox / (FH2 * FH1);
          const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
          const int c = cc * CK + c4 * 4;
          const bool ok = (k0 + kk < NIT) & (f < FHV * C4) & (gz >= 0) & (gz < D0) & (gy >= 0) & (gy < D1) & (gx >= 0) &
                          (gx < D2);
          const size_t off =
              ok ? ((((size_t)(gz * is + io0) * (D1 * is) + (gy * is + io1)) * (D2 * is) + (gx * is + io2)) * Cin + c) : 0;
          float4 v;
          if constexpr (CK != 8) {  // Cin % CK == 0: aligned float4, channel range always valid
            v = ld4(in + off);
            if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
          } else {
            const float* src = in + off;
            v.x = (ok && c + 0 < Cin) ? src[0] : 0.f;
            v.y = (ok && c + 1 < Cin) ? src[(c + 1 < Cin) ? 1 : 0] : 0.f;
            v.z = (ok && c + 2 < Cin) ? src[(c + 2 < Cin) ? 2 : 0] : 0.f;
            v.w = (ok && c + 3 < Cin) ? src[(c + 3 < Cin) ? 3 : 0] : 0.f;
          }
          stg[kk] = v;
        }
#pragma unroll
        for (int kk = 0; kk < SB; ++kk) {
          const int f = tid + (k0 + kk) * 256;
          const int vox = f / C4, c4 = f - vox * C4;
          if (k0 + kk < NIT && f < FHV * C4) *reinterpret_cast<float4*>(&lds[vox * CKP + c4 * 4]) = stg[kk];
        }
      }
    }
    __syncthreads();

    const float* wc = wl + (size_t)cc * 27 * NCG * NT * 128 + (size_t)par * ext.wstride;
    const float* wvc = wp + ext.valu_off + (size_t)par * ext.wstride + (size_t)cc * 27 * CK * (NV > 0 ? NV : 1);
    // Software pipeline, pinned with sched_barrier so that hipcc cannot sink the prefetches next to their uses:
    //   B fragments of the next active tap are requested at the top of a tap (one L2 round trip hidden behind
    //   16*NCG*NT MFMAs), A fragments of step (t,g)+1 are read from LDS before the MFMAs of step (t,g).
    int tap = __builtin_ctz(tapmask);
    auto tap_off = [](int t) { return (((t / 9) * FH1 + (t / 3) % 3) * FH2 + t % 3) * CKP; };
    float2 bcur[NCG][NT], bnext[NCG][NT];
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
      for (int n = 0; n < NT; ++n)
        bcur[g][n] = *reinterpret_cast<const float2*>(wc + (size_t)tap * NCG * NT * 128 + (g * NT + n) * 128);
    float2 acur[MT], anext[MT];
    float4 xcur[2], xnext[2];
    {
      const int o = tap_off(tap);
#pragma unroll
      for (int m = 0; m < MT; ++m) acur[m] = *reinterpret_cast<const float2*>(&lds[a_base[m] + o]);
      if constexpr (NV > 0) {
        xcur[0] = *reinterpret_cast<const float4*>(&lds[vbase + o]);
        xcur[1] = *reinterpret_cast<const float4*>(&lds[vbase + o + 4]);
      }
    }
    while (tap < 27) {
      const uint32_t rem = (tap + 1 < 27) ? (tapmask >> (tap + 1)) : 0u;
      const int tn = rem ? tap + 1 + __builtin_ctz(rem) : 27;
      if (tn < 27 && !(dbg & 1)) {
        const float* wn = wc + (size_t)tn * NCG * NT * 128;
#pragma unroll
        for (int g = 0; g < NCG; ++g)
#pragma unroll
          for (int n = 0; n < NT; ++n) bnext[g][n] = *reinterpret_cast<const float2*>(wn + (g * NT + n) * 128);
      }
      const int toff = tap_off(tap);
      const int toff_n = tap_off(tn < 27 ? tn : tap);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < NCG; ++g) {
        const int noff = (g + 1 < NCG) ? toff + (g + 1) * 8 : toff_n;
        if (!(dbg & 2))
#pragma unroll
          for (int m = 0; m < MT; ++m) anext[m] = *reinterpret_cast<const float2*>(&lds[a_base[m] + noff]);
        if constexpr (NV > 0) {
          xnext[0] = *reinterpret_cast<const float4*>(&lds[vbase + noff]);
          xnext[1] = *reinterpret_cast<const float4*>(&lds[vbase + noff + 4]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NV > 0) {  // 8 ci x NV co v_fma per lane; weights are wave-uniform -> scalar loads
          const float* wv = wvc + ((size_t)tap * CK + g * 8) * NV;
          const float xs[8] = {xcur[0].x, xcur[0].y, xcur[0].z, xcur[0].w, xcur[1].x, xcur[1].y, xcur[1].z, xcur[1].w};
#pragma unroll
          for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int v = 0; v < NV; ++v) accv[v] = fmaf(xs[c], wv[c * NV + v], accv[v]);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[m].x, bcur[g][n].x, acc[m][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[m].y, bcur[g][n].y, acc[m][n], 0, 0, 0);
        if constexpr (NV > 0) {  // interleave: one MFMA, then its share of the packed v_fma (issued in the MFMA's shadow)
#pragma unroll
          for (int q = 0; q < 2 * MT * NT; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, (8 * NV / 2) / (2 * MT * NT), 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m) acur[m] = anext[m];
        if constexpr (NV > 0) {
          xcur[0] = xnext[0];
          xcur[1] = xnext[1];
        }
      }
#pragma unroll
      for (int g = 0; g < NCG; ++g)
#pragma unroll
        for (int n = 0; n < NT; ++n) bcur[g][n] = bnext[g][n];
      tap = tn;
    }
  }

  // ---- epilogue
  const int gz = z0 + wave;
  if (gz < D0 && !(dbg & 8)) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int gy = y0 + m;
      if (gy >= D1) continue;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int co = (nc * NT + n) * 16 + li;
        if (co >= Cout - NV) continue;
        const float bv = (!KSPLIT && bias) ? bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gx = x0 + kq * 4 + r;
          if (gx < D2) {
            const int os = (ext.mode == 1) ? 2 : 1;
            const size_t oidx = (((size_t)(gz * os + ((opar >> 2) & 1)) * (D1 * os) + (gy * os + ((opar >> 1) & 1))) *
                                     (D2 * os) + (gx * os + (opar & 1))) * Cout + co;
            if constexpr (KSPLIT) {
              atomicAdd(out + oidx, acc[m][n][r]);
            } else {
              float v = acc[m][n][r] + bv;
              if (act == 2) v *= elu_dy(ext.addend[oidx]);
              else if (ext.addend) v += ext.addend[oidx];
              if (act == 1) v = elu_f(v);
              out[oidx] = v;
            }
          }
        }
      }
    }
  }
  if constexpr (NV > 0) {  // the lane's own voxel, channels Cout-NV .. Cout-1
    const int vy = y0 + (lane >> 4), vx = x0 + (lane & 15);
    if (gz < D0 && vy < D1 && vx < D2 && !(dbg & 8)) {
      const int os = (ext.mode == 1) ? 2 : 1;
      const size_t oidx = (((size_t)(gz * os + ((opar >> 2) & 1)) * (D1 * os) + (vy * os + ((opar >> 1) & 1))) * (D2 * os) +
                           (vx * os + (opar & 1))) * Cout + (Cout - NV);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        float r = accv[v] + (bias ? bias[Cout - NV + v] : 0.f);
        if (ext.addend) r += ext.addend[oidx + v];
        if (act == 1) r = elu_f(r);
        out[oidx + v] = r;
      }
    }
  }
}

// ---- persistent variant for the large levels (CK = 24, 4x4x16 tiles) ----------------------------------------
// gridDim.x workgroups (2 per CU) walk the (tile, channel-chunk) items round-robin.  While the 27 taps of item i
// run on the matrix cores, the halo tile of item i+1 is fetched into registers: one 16-byte global load per tap,
// interleaved with the B-fragment loads, so its HBM latency hides behind ~2 taps (~100 MFMAs) of compute.
// Only the LDS store phase (2 barriers) separates two items; the output stores of an item overlap the next one.
template <int NT>
__global__ __launch_bounds__(256, 2) void conv3d_fwd_persist_kernel(const float* __restrict__ in,
                                                                    const float* __restrict__ wp,
                                                                    const float* __restrict__ bias,
                                                                    float* __restrict__ out, int D0, int D1, int D2,
                                                                    int Cin, int Cout, int ncc, int tiles1, int tiles2,
                                                                    int ntiles, int act, const float* addend) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CK = 24, MT = 4;
  constexpr int FT1 = MT, FH1 = MT + 2, FHV = FH0 * FH1 * FH2;
  constexpr int CKP = CK + 4, NCG = CK / 8, C4 = CK / 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int nc = blockIdx.y;
  const int nitems = ntiles * ncc;
  // XCD-aware order: in every round of gridDim.x items, workgroup b (XCD b%8) takes position (b%8)*(G/8) + b/8,
  // so each XCD's L2 serves a contiguous run of neighbouring tiles.  gridDim.x is a multiple of 8.
  const int G = gridDim.x;
  const int my_pos = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);

  int a_base[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) a_base[m] = ((wave * FH1 + m) * FH2 + li) * CKP + 2 * kq;
  const float* wl = wp + (size_t)nc * ncc * 27 * NCG * NT * 128 + lane * 2;

  // ---- VALU-lean addressing.  On CDNA4 every vector-ALU instruction takes ~3-4 cycles out of the SIMD's MFMA issue
  // (tools/ubench/mfma_issue.hip), so the per-item address arithmetic is moved to the scalar unit:
  //  * halo staging: in every z-plane of the 6x6x18 halo tile, thread t owns the same <= 3 (y, x, channel-quad) columns
  //    j = t + 256 i of the plane's 648 float4.  Their byte offsets relative to the tile origin and their LDS addresses
  //    are computed ONCE per kernel; per item only `offset + scalar` and a bit-mask validity test remain.  The loads are
  //    raw buffer loads: soffset carries the z-plane (scalar), out-of-volume elements get an offset beyond
  //    num_records and come back as zeros (hardware range check) -- no per-element clamping or selects.
  //  * B fragments: buffer loads with voffset = lane*8, soffset = scalar (chunk, tap), immediate = (g, n).
  constexpr int PLANE4 = FH1 * FH2 * C4;  // 648 float4 per halo z-plane
  constexpr int NJ = (PLANE4 + 255) / 256;  // 3 columns per thread
  constexpr int NLD = NJ * FH0;             // 18 halo loads per item
  static_assert(NLD <= 27, "one halo load per tap");
  constexpr uint32_t OOB = 0x80000000u;  // > num_records (launcher guarantees tensor bytes < 2^31)
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)((int64_t)D0 * D1 * D2 * Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(wp), 0, (int)((int64_t)gridDim.y * ncc * 27 * NCG * NT * 128 * 4), 0x00020000);
  int rel[NJ], ldsa[NJ];
  uint32_t cmask[NJ];  // one-hot (1 << hy) | (1 << (8 + hx)); all ones for j >= 648
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    const int j = tid + 256 * i;
    const int hy = j / (FH2 * C4), r = j - hy * (FH2 * C4), hx = r / C4, c4 = r - hx * C4;
    rel[i] = ((hy * D2 + hx) * Cin + c4 * 4) * 4;
    ldsa[i] = ((hy * FH2 + hx) * CKP + c4 * 4);
    cmask[i] = j < PLANE4 ? ((1u << hy) | (1u << (8 + hx))) : 0xFFFFFFFFu;
  }
  const int plane_bytes = D1 * D2 * Cin * 4;
  // per item: voff[i] = byte offset of column i inside plane z (or OOB)
  auto item_offsets = [&](int y0, int x0, int cc, uint32_t (&voff)[NJ]) {
    // invalid local rows/columns of this tile (scalar): hy valid iff 0 <= y0-1+hy < D1
    uint32_t bad = 0x80000000u;  // bit 31: always-invalid marker for j >= 648
#pragma unroll
    for (int h = 0; h < FH1; ++h) bad |= ((unsigned)(y0 - 1 + h) >= (unsigned)D1) ? (1u << h) : 0u;
#pragma unroll
    for (int h = 0; h < FH2; ++h) bad |= ((unsigned)(x0 - 1 + h) >= (unsigned)D2) ? (1u << (8 + h)) : 0u;
    const int yx = (((y0 - 1) * D2 + (x0 - 1)) * Cin + cc * CK) * 4;
#pragma unroll
    for (int i = 0; i < NJ; ++i) voff[i] = (cmask[i] & bad) ? OOB : (uint32_t)(rel[i] + yx);
  };
  // halo load k = plane * NJ + i
  auto halo_load = [&](int k, int z0, const uint32_t (&voff)[NJ]) -> float4 {
    const int hz = k / NJ, i = k - hz * NJ;
    const int gz = z0 - 1 + hz;
    const bool pv = (unsigned)gz < (unsigned)D0;  // scalar
    const uint32_t vo = pv ? voff[i] : OOB;
    const int so = pv ? gz * plane_bytes : 0;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)vo, so, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  };
  auto tile_origin = [&](int t, int& z0, int& y0, int& x0) {
    const int t2 = t % tiles2, t1 = (t / tiles2) % tiles1, t0 = t / (tiles2 * tiles1);
    z0 = t0 * FT0;
    y0 = t1 * FT1;
    x0 = t2 * FT2;
  };
  auto bload = [&](int soff, int idx) -> float2 {  // idx = g*NT + n; immediates stay below 4096
    const int hi = idx >> 2, lo = idx & 3;
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rw, lane * 8 + lo * 512, soff + hi * 2048, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
  };

  f32x4 acc[MT][NT];
  float4 stg[NLD];
  // items of this workgroup: its tiles (my_pos, my_pos + G, ...), and for each tile ALL channel chunks in order, so that
  // the accumulators of a tile stay in this workgroup's registers
  int tile = my_pos;
  if (tile >= ntiles) return;
  (void)nitems;
  int z0, y0, x0, cc = 0;
  tile_origin(tile, z0, y0, x0);
  {
    uint32_t voff[NJ];
    item_offsets(y0, x0, cc, voff);
#pragma unroll
    for (int k = 0; k < NLD; ++k) stg[k] = halo_load(k, z0, voff);
  }

  while (true) {
    const int ncc_ = (cc + 1 < ncc) ? cc + 1 : 0;
    const int ntile = (cc + 1 < ncc) ? tile : tile + G;
    const bool has_next = ntile < ntiles;
    int nz0 = z0, ny0 = y0, nx0 = x0;
    if (has_next && ntile != tile) tile_origin(ntile, nz0, ny0, nx0);
    uint32_t nvoff[NJ];
    item_offsets(ny0, nx0, ncc_, nvoff);
    if (cc == 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();  // every wave has finished reading the previous item's tile
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      if (i < NJ - 1 || tid + 256 * i < PLANE4) {
#pragma unroll
        for (int hz = 0; hz < FH0; ++hz)
          *reinterpret_cast<float4*>(&lds[ldsa[i] + hz * (FH1 * FH2 * CKP)]) = stg[hz * NJ + i];
      }
    }
    __syncthreads();

    const int wsoff = ((nc * ncc + cc) * 27) * (NCG * NT * 512);  // bytes
    // Register ping-pong, statically indexed (the tap loop is fully unrolled): fragments are prefetched into the set the
    // MFMAs are NOT reading, so no v_mov copies follow the MFMAs.
    float2 bb[2][NCG][NT];
    float2 aa[2][MT];
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
      for (int n = 0; n < NT; ++n) bb[0][g][n] = bload(wsoff, g * NT + n);
#pragma unroll
    for (int m = 0; m < MT; ++m) aa[0][m] = *reinterpret_cast<const float2*>(&lds[a_base[m]]);

#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      if (tap + 1 < 27) {
#pragma unroll
        for (int g = 0; g < NCG; ++g)
#pragma unroll
          for (int n = 0; n < NT; ++n) bb[(tap + 1) & 1][g][n] = bload(wsoff + (tap + 1) * (NCG * NT * 512), g * NT + n);
      }
      if (tap < NLD) {
        if (has_next) stg[tap] = halo_load(tap, nz0, nvoff);  // wave-uniform branch
      }
      const int tn = tap + 1 < 27 ? tap + 1 : 26;
      const int toff = (((tap / 9) * FH1 + (tap / 3) % 3) * FH2 + tap % 3) * CKP;
      const int toff_n = (((tn / 9) * FH1 + (tn / 3) % 3) * FH2 + tn % 3) * CKP;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < NCG; ++g) {
        const int st = tap * NCG + g;  // compile-time after unrolling
        const int noff = (g + 1 < NCG) ? toff + (g + 1) * 8 : toff_n;
#pragma unroll
        for (int m = 0; m < MT; ++m) aa[(st + 1) & 1][m] = *reinterpret_cast<const float2*>(&lds[a_base[m] + noff]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[st & 1][m].x, bb[tap & 1][g][n].x, acc[m][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[st & 1][m].y, bb[tap & 1][g][n].y, acc[m][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    if (cc == ncc - 1) {  // epilogue of this tile (Cout % 4 == 0 is guaranteed by the launcher)
      __syncthreads();
      const int gz = z0 + wave;
      if (gz < D0)
        store_tile_rows<NT, MT>(acc, lds + wave * (16 * NT * 16), out, bias, addend, act, nc, gz, y0, x0, D1, D2, Cout, lane);
    }
    if (!has_next) break;
    tile = ntile;
    z0 = nz0;
    y0 = ny0;
    x0 = nx0;
    cc = ncc_;
  }
}

// ---- persistent forward kernel for Cout = 24 on v_mfma_f32_4x4x1_16B_f32 ---------------------------------------
// The 16x16x4 MFMA pads 24 output channels to 32 columns (25 % of the matrix work wasted on every level-0 layer).
// The 4x4x1 MFMA runs 16 independent 4x4 outer products per instruction at the same FLOP rate (512 FLOP / 8 cycles,
// tools/ubench/mfma_4x4.hip: 136-147 TF) and its CBSZ/ABID fields broadcast ONE block of the A operand to all 16
// blocks.  Mapping: A = weights (rows = 4 output channels), B = activations (block b, column j = voxel 4b+j = lane),
// D[i] of lane l = out[voxel l][4g+i].  A weight register therefore holds 16 different channel groups (selected by
// ABID) and is loaded once per wave, the lane's own voxel supplies B from the LDS halo tile with one ds_read_b128 per
// 4 input channels, and every lane ends up with the 24 output channels of its voxel in 24 accumulator registers
// (epilogue = 6 contiguous float4 stores per lane, no LDS transpose).  24 = 6 groups of 4: no padding.
// Tile, halo staging and item order are those of conv3d_fwd_persist_kernel.
__global__ __launch_bounds__(256, 2) void conv3d_fwd_p4_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                               const float* __restrict__ bias, float* __restrict__ out,
                                                               int D0, int D1, int D2, int Cin, int ncc, int tiles1,
                                                               int tiles2, int ntiles, int act, const float* addend,
                                                               float* __restrict__ stats_partial) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CK = 24, MT = 4, Cout = 24;
  constexpr int FT1 = MT, FH1 = MT + 2;
  constexpr int CKP = CK + 4, C4 = CK / 4;
  const int dbg = act >> 8;  // timing experiments (synthsr_conv3d_set_option 1): 64 no LDS restage, 128 no epilogue
  act &= 0xff;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = gridDim.x;
  const int my_pos = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int vy = lane >> 4, vx = lane & 15;
  const int xbase = ((wave * FH1 + vy) * FH2 + vx) * CKP;

  constexpr int PLANE4 = FH1 * FH2 * C4, NJ = (PLANE4 + 255) / 256, NLD = NJ * FH0;
  constexpr uint32_t OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)((int64_t)D0 * D1 * D2 * Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, ncc * 27 * 9 * 256, 0x00020000);
  int rel[NJ], ldsa[NJ];
  uint32_t cmask[NJ];
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    const int j = tid + 256 * i;
    const int hy = j / (FH2 * C4), r = j - hy * (FH2 * C4), hx = r / C4, c4 = r - hx * C4;
    rel[i] = ((hy * D2 + hx) * Cin + c4 * 4) * 4;
    ldsa[i] = ((hy * FH2 + hx) * CKP + c4 * 4);
    cmask[i] = j < PLANE4 ? ((1u << hy) | (1u << (8 + hx))) : 0xFFFFFFFFu;
  }
  const int plane_bytes = D1 * D2 * Cin * 4;
  auto item_offsets = [&](int y0, int x0, int cc, uint32_t (&voff)[NJ]) {
    uint32_t bad = 0x80000000u;
#pragma unroll
    for (int h = 0; h < FH1; ++h) bad |= ((unsigned)(y0 - 1 + h) >= (unsigned)D1) ? (1u << h) : 0u;
#pragma unroll
    for (int h = 0; h < FH2; ++h) bad |= ((unsigned)(x0 - 1 + h) >= (unsigned)D2) ? (1u << (8 + h)) : 0u;
    const int yx = (((y0 - 1) * D2 + (x0 - 1)) * Cin + cc * CK) * 4;
#pragma unroll
    for (int i = 0; i < NJ; ++i) voff[i] = (cmask[i] & bad) ? OOB : (uint32_t)(rel[i] + yx);
  };
  auto halo_load = [&](int k, int z0, const uint32_t (&voff)[NJ]) -> float4 {
    const int hz = k / NJ, i = k - hz * NJ;
    const int gz = z0 - 1 + hz;
    const bool pv = (unsigned)gz < (unsigned)D0;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)(pv ? voff[i] : OOB), pv ? gz * plane_bytes : 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  };
  auto tile_origin = [&](int t, int& z0, int& y0, int& x0) {
    const int t2 = t % tiles2, t1 = (t / tiles2) % tiles1, t0 = t / (tiles2 * tiles1);
    z0 = t0 * FT0;
    y0 = t1 * FT1;
    x0 = t2 * FT2;
  };
  auto wload = [&](int soff, int r) -> float {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw, lane * 4 + r * 256, soff, 0));
  };

  f32x4 acc[6];
  // results are stored from these registers, which nothing else touches until the next epilogue: re-zeroing `acc` (or
  // reusing temporaries) right after a store would make the wave wait for the store's completion (vmcnt on gfx9)
  float4 outreg[6];
  // optional BatchNorm statistics of the output: per-lane partial sums / sums of squares over the lane's voxels, reduced
  // once per workgroup into stats_partial[blockIdx.x][48] (no global atomics: thousands of same-address atomics
  // serialise); saves the separate bn_stats pass over the 393 MB activation
  float ps[24], pq[24];
#pragma unroll
  for (int c = 0; c < 24; ++c) ps[c] = pq[c] = 0.f;
  float bv[24];  // bias, wave-uniform (scalar registers): loaded once, not per tile
#pragma unroll
  for (int c = 0; c < 24; ++c) bv[c] = bias ? bias[c] : 0.f;
  float4 stg[NLD];
  int tile = my_pos;
  if (tile >= ntiles) {
    if (stats_partial && tid < 48) stats_partial[(size_t)blockIdx.x * 48 + tid] = 0.f;
    return;
  }
  int z0, y0, x0, cc = 0;
  tile_origin(tile, z0, y0, x0);
  {
    uint32_t voff[NJ];
    item_offsets(y0, x0, cc, voff);
#pragma unroll
    for (int k = 0; k < NLD; ++k) stg[k] = halo_load(k, z0, voff);
  }
  // weight ring: step p of an item uses wr[p % 3]; steps are requested two ahead, and since 81 = 0 (mod 3) the ring
  // simply continues into the next item (whose first two steps are requested by the last two steps of this one)
  constexpr int NP = 27 * 3;
  float wr[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    wr[0][r] = wload(0, r);
    wr[1][r] = wload(768, r);
  }

  while (true) {
    const int ncc_ = (cc + 1 < ncc) ? cc + 1 : 0;
    const int ntile = (cc + 1 < ncc) ? tile : tile + G;
    const bool has_next = ntile < ntiles;
    int nz0 = z0, ny0 = y0, nx0 = x0;
    if (has_next && ntile != tile) tile_origin(ntile, nz0, ny0, nx0);
    uint32_t nvoff[NJ];
    item_offsets(ny0, nx0, ncc_, nvoff);
    if (cc == 0) {
#pragma unroll
      for (int g = 0; g < 6; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (!(dbg & 64)) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      if (i < NJ - 1 || tid + 256 * i < PLANE4) {
#pragma unroll
        for (int hz = 0; hz < FH0; ++hz)
          *reinterpret_cast<float4*>(&lds[ldsa[i] + hz * (FH1 * FH2 * CKP)]) = stg[hz * NJ + i];
      }
    }
    __syncthreads();
    }

    // 81 channel-octet steps (tap, qp); weights of step p live in wr[p % 3] and are requested two steps ahead, the
    // activations of quad step s (two per octet) in xq[s & 1], read one step ahead
    const int wsoff = cc * NP * 768, wsoff_n = ncc_ * NP * 768;
    float4 xq[2];
    xq[0] = *reinterpret_cast<const float4*>(&lds[xbase]);
    sfor<0, NP>([&](auto P) {
      constexpr int p = decltype(P)::value, tap = p / 3, qp = p % 3;
      constexpr int toff = (((tap / 9) * FH1 + (tap / 3) % 3) * FH2 + tap % 3) * CKP;
      if constexpr (p + 2 < NP) {
#pragma unroll
        for (int r = 0; r < 3; ++r) wr[(p + 2) % 3][r] = wload(wsoff + (p + 2) * 768, r);
      } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) wr[(p + 2) % 3][r] = wload(wsoff_n + (p + 2 - NP) * 768, r);
      }
      if constexpr (qp == 0 && tap < NLD) {
        if (has_next) stg[tap] = halo_load(tap, nz0, nvoff);  // wave-uniform branch
      }
      sfor<0, 2>([&](auto H) {
        constexpr int h = decltype(H)::value, s = p * 2 + h;
        // next quad step: (p, 1) after (p, 0); (p + 1, 0) after (p, 1)
        constexpr int pn = h == 0 ? p : (p + 1 < NP ? p + 1 : p), hn = h == 0 ? 1 : 0;
        constexpr int tn = pn / 3, qn = (pn % 3) * 2 + hn;
        constexpr int noff = (((tn / 9) * FH1 + (tn / 3) % 3) * FH2 + tn % 3) * CKP + qn * 4;
        (void)toff;
        xq[(s + 1) & 1] = *reinterpret_cast<const float4*>(&lds[xbase + noff]);
        __builtin_amdgcn_sched_barrier(0);
        const float xs[4] = {xq[s & 1].x, xq[s & 1].y, xq[s & 1].z, xq[s & 1].w};
        sfor<0, 24>([&](auto GI) {
          constexpr int gi = decltype(GI)::value, kk = gi / 6, g = gi % 6, GG = h * 24 + gi;
          acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[p % 3][GG / 16], xs[kk], acc[g], 4, GG % 16, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    });

    if (cc == ncc - 1 && !(dbg & 128)) {  // epilogue: the lane's voxel, 24 channels = 6 float4
      const int gz = z0 + wave, gy = y0 + vy, gx = x0 + vx;
      if (gz < D0 && gy < D1 && gx < D2) {
        const size_t o = (((size_t)gz * D1 + gy) * D2 + gx) * Cout;
#pragma unroll
        for (int g = 0; g < 6; ++g) {
          float4 v = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
          v.x += bv[4 * g];
          v.y += bv[4 * g + 1];
          v.z += bv[4 * g + 2];
          v.w += bv[4 * g + 3];
          if (act == 2) {  // fused ELU backward of the producing layer (addend = its output)
            const float4 a = *reinterpret_cast<const float4*>(addend + o + 4 * g);
            v.x *= elu_dy(a.x);
            v.y *= elu_dy(a.y);
            v.z *= elu_dy(a.z);
            v.w *= elu_dy(a.w);
          } else if (addend) {  // may alias out: same element read and written by this lane
            const float4 a = *reinterpret_cast<const float4*>(addend + o + 4 * g);
            v.x += a.x;
            v.y += a.y;
            v.z += a.z;
            v.w += a.w;
          }
          if (act == 1) {
            v.x = elu_f(v.x);
            v.y = elu_f(v.y);
            v.z = elu_f(v.z);
            v.w = elu_f(v.w);
          }
          outreg[g] = v;
          if (stats_partial) {  // wave-uniform
            ps[4 * g] += v.x;
            ps[4 * g + 1] += v.y;
            ps[4 * g + 2] += v.z;
            ps[4 * g + 3] += v.w;
            pq[4 * g] = fmaf(v.x, v.x, pq[4 * g]);
            pq[4 * g + 1] = fmaf(v.y, v.y, pq[4 * g + 1]);
            pq[4 * g + 2] = fmaf(v.z, v.z, pq[4 * g + 2]);
            pq[4 * g + 3] = fmaf(v.w, v.w, pq[4 * g + 3]);
          }
        }
#pragma unroll
        for (int g = 0; g < 6; ++g) {
          asm volatile("" : "+v"(outreg[g].x), "+v"(outreg[g].y), "+v"(outreg[g].z), "+v"(outreg[g].w));  // own registers
          *reinterpret_cast<float4*>(out + o + 4 * g) = outreg[g];
        }
      }
    }
    if (!has_next) break;
    tile = ntile;
    z0 = nz0;
    y0 = ny0;
    x0 = nx0;
    cc = ncc_;
  }
  if (stats_partial) {  // wave butterfly -> LDS [4 waves][48] -> one row of the partial buffer per workgroup
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 24; ++c) {
      float a = ps[c], q = pq[c];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        a += __shfl_xor(a, m, 64);
        q += __shfl_xor(q, m, 64);
      }
      if (lane == 0) {
        lds[wave * 48 + c] = a;
        lds[wave * 48 + 24 + c] = q;
      }
    }
    __syncthreads();
    if (tid < 48) stats_partial[(size_t)blockIdx.x * 48 + tid] = lds[tid] + lds[48 + tid] + lds[96 + tid] + lds[144 + tid];
  }
}

// ---- forward parity convs of a folded decoder conv with Cout = 24 (nearest-upsample folding), 4x4x1 MFMA ---------------
// conv3d_fwd_p4_kernel with the 2x2x2 parity windows: lane = LOW-RES voxel.  One staged low-res halo chunk serves the 4
// output parities (py, px) of one pz (4 x 6 accumulators): 4 stagings per tile instead of the 16 of the per-parity
// launch, no padding of the 24 output channels.  items = (tile, pz, channel chunk); each lane finally stores its 4
// hi-res voxels (2z+pz, 2y+py, 2x+px), 24 channels each.  Weights: p4 layout of the parity-combined 27-slot kernels.
__global__ __launch_bounds__(256, 2) void conv3d_up_fwd_p4_kernel(const float* __restrict__ in,
                                                                  const float* __restrict__ wp,
                                                                  const float* __restrict__ bias, float* __restrict__ out,
                                                                  int D0, int D1, int D2, int Cin, int ncc, int tiles1,
                                                                  int tiles2, int ntiles, int act, int64_t wstride,
                                                                  const float* addend) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CK = 24, MT = 4, Cout = 24;
  constexpr int FT1 = MT, FH1 = MT + 2, CKP = CK + 4, C4 = CK / 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = gridDim.x;
  const int my_pos = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int vy = lane >> 4, vx = lane & 15;
  const int xbase = ((wave * FH1 + vy) * FH2 + vx) * CKP;
  constexpr int PLANE4 = FH1 * FH2 * C4, NJ = (PLANE4 + 255) / 256, NLD = NJ * FH0;
  constexpr uint32_t OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)((int64_t)D0 * D1 * D2 * Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, 0x7FFFFFF0, 0x00020000);
  int rel[NJ], ldsa[NJ];
  uint32_t cmask[NJ];
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    const int j = tid + 256 * i;
    const int hy = j / (FH2 * C4), r = j - hy * (FH2 * C4), hx = r / C4, c4 = r - hx * C4;
    rel[i] = ((hy * D2 + hx) * Cin + c4 * 4) * 4;
    ldsa[i] = ((hy * FH2 + hx) * CKP + c4 * 4);
    cmask[i] = j < PLANE4 ? ((1u << hy) | (1u << (8 + hx))) : 0xFFFFFFFFu;
  }
  const int plane_bytes = D1 * D2 * Cin * 4;
  float4 stg[NLD];
  auto tile_origin = [&](int t, int& z0, int& y0, int& x0) {
    const int t2 = t % tiles2, t1 = (t / tiles2) % tiles1, t0 = t / (tiles2 * tiles1);
    z0 = t0 * FT0;
    y0 = t1 * FT1;
    x0 = t2 * FT2;
  };
  auto halo_loads = [&](int z0, int y0, int x0, int cc) {
    uint32_t bad = 0x80000000u;
#pragma unroll
    for (int h = 0; h < FH1; ++h) bad |= ((unsigned)(y0 - 1 + h) >= (unsigned)D1) ? (1u << h) : 0u;
#pragma unroll
    for (int h = 0; h < FH2; ++h) bad |= ((unsigned)(x0 - 1 + h) >= (unsigned)D2) ? (1u << (8 + h)) : 0u;
    const int yx = (((y0 - 1) * D2 + (x0 - 1)) * Cin + cc * CK) * 4;
    uint32_t voff[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) voff[i] = (cmask[i] & bad) ? OOB : (uint32_t)(rel[i] + yx);
#pragma unroll
    for (int hz = 0; hz < FH0; ++hz) {
      const int gz = z0 - 1 + hz;
      const bool pv = (unsigned)gz < (unsigned)D0;
#pragma unroll
      for (int i = 0; i < NJ; ++i) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)(pv ? voff[i] : OOB), pv ? gz * plane_bytes : 0, 0);
        stg[hz * NJ + i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
    }
  };
  auto wload = [&](int soff, int r) -> float {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw, lane * 4 + r * 256, soff, 0));
  };
  float bv[24];
#pragma unroll
  for (int c = 0; c < 24; ++c) bv[c] = bias ? bias[c] : 0.f;

  f32x4 acc[4][6];
  int tile = my_pos;
  if (tile >= ntiles) return;
  int z0, y0, x0;
  tile_origin(tile, z0, y0, x0);
  halo_loads(z0, y0, x0, 0);
  const int nsub = 2 * ncc;  // items of one tile: (pz, cc)
  int sub = 0;
  while (true) {
    const int pz = sub / ncc, cc = sub - pz * ncc;
    // next item
    const int nsub_i = (sub + 1 < nsub) ? sub + 1 : 0;
    const int ntile = (sub + 1 < nsub) ? tile : tile + G;
    const bool has_next = ntile < ntiles;
    int nz0 = z0, ny0 = y0, nx0 = x0;
    if (has_next && ntile != tile) tile_origin(ntile, nz0, ny0, nx0);
    if (cc == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int g = 0; g < 6; ++g) acc[q][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      if (i < NJ - 1 || tid + 256 * i < PLANE4) {
#pragma unroll
        for (int hz = 0; hz < FH0; ++hz)
          *reinterpret_cast<float4*>(&lds[ldsa[i] + hz * (FH1 * FH2 * CKP)]) = stg[hz * NJ + i];
      }
    }
    __syncthreads();
    if (has_next) halo_loads(nz0, ny0, nx0, nsub_i % ncc);

    // 96 steps = 4 parities (py, px) x 8 window taps x 3 channel octets; window position inside the 3x3x3 stencil: shift =
    // parity (see up_tapmask).  Weights of step p in wr[p % 3] (requested two steps ahead), activations of quad step s
    // in xq[s & 1] (read one step ahead) -- as in conv3d_fwd_p4_kernel.
    constexpr int NP = 4 * 8 * 3;
    const int xb = xbase + pz * (FH1 * FH2 * CKP);  // z shift of the window (runtime); y / x shifts are static
    auto wsoff = [&](int p) {  // scalar byte offset of the weights of step p
      const int q = p / 24, ti = (p / 3) % 8, qp = p % 3;
      const int tap = ((pz + ((ti >> 2) & 1)) * 3 + ((q >> 1) + ((ti >> 1) & 1))) * 3 + ((q & 1) + (ti & 1));
      return (int)((int64_t)(pz * 4 + q) * wstride * 4) + cc * (27 * 9 * 256) + (tap * 3 + qp) * 768;
    };
    float wr[3][3];
    float4 xq[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      wr[0][r] = wload(wsoff(0), r);
      wr[1][r] = wload(wsoff(1), r);
    }
    xq[0] = *reinterpret_cast<const float4*>(&lds[xb]);
    sfor<0, NP>([&](auto P) {
      constexpr int p = decltype(P)::value, q = p / 24;
      if constexpr (p + 2 < NP) {
#pragma unroll
        for (int r = 0; r < 3; ++r) wr[(p + 2) % 3][r] = wload(wsoff(p + 2), r);
      }
      sfor<0, 2>([&](auto H) {
        constexpr int h = decltype(H)::value, sidx = p * 2 + h;
        constexpr int pn = h == 0 ? p : (p + 1 < NP ? p + 1 : p), hn = h == 0 ? 1 : 0;
        constexpr int qn = pn / 24, tn = (pn / 3) % 8, c4n = (pn % 3) * 2 + hn;
        constexpr int noff = ((((tn >> 2) & 1) * FH1 + (qn >> 1) + ((tn >> 1) & 1)) * FH2 + (qn & 1) + (tn & 1)) * CKP + c4n * 4;
        xq[(sidx + 1) & 1] = *reinterpret_cast<const float4*>(&lds[xb + noff]);
        __builtin_amdgcn_sched_barrier(0);
        const float xs[4] = {xq[sidx & 1].x, xq[sidx & 1].y, xq[sidx & 1].z, xq[sidx & 1].w};
        sfor<0, 24>([&](auto GI) {
          constexpr int gi = decltype(GI)::value, kk = gi / 6, g = gi % 6, GG = h * 24 + gi;
          acc[q][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[p % 3][GG / 16], xs[kk], acc[q][g], 4, GG % 16, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    });

    if (cc == ncc - 1) {  // epilogue of (tile, pz): the lane's 4 hi-res voxels
      const int gz = z0 + wave, gy = y0 + vy, gx = x0 + vx;
      if (gz < D0 && gy < D1 && gx < D2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const size_t o = (((size_t)(2 * gz + pz) * (2 * D1) + (2 * gy + (q >> 1))) * (2 * D2) + (2 * gx + (q & 1))) * Cout;
#pragma unroll
          for (int g = 0; g < 6; ++g) {
            float4 v = make_float4(acc[q][g][0] + bv[4 * g], acc[q][g][1] + bv[4 * g + 1], acc[q][g][2] + bv[4 * g + 2],
                                   acc[q][g][3] + bv[4 * g + 3]);
            if (addend) {  // indexed like out; may alias it
              const float4 a = *reinterpret_cast<const float4*>(addend + o + 4 * g);
              v.x += a.x;
              v.y += a.y;
              v.z += a.z;
              v.w += a.w;
            }
            if (act == 1) {
              v.x = elu_f(v.x);
              v.y = elu_f(v.y);
              v.z = elu_f(v.z);
              v.w = elu_f(v.w);
            }
            *reinterpret_cast<float4*>(out + o + 4 * g) = v;
          }
        }
      }
    }
    if (!has_next) break;
    tile = ntile;
    z0 = nz0;
    y0 = ny0;
    x0 = nx0;
    sub = nsub_i;
  }
}

// ---- first layer (Cin <= 2, Cout = 24) on the 4x4x1 MFMA: K = 27*Cin, all weights resident in <= 21 registers ------
// The generic path pads Cin = 2 to an 8-channel chunk (4x the matrix work) and is bound by everything but memory;
// this kernel is bound by the 160^3 x 24 output write.  Lane = voxel as in conv3d_fwd_p4_kernel; the halo tile is
// [648 voxels][Cin] in LDS (5 KB).
template <int CIN>
__global__ __launch_bounds__(256, 2) void conv3d_fwd_c2_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                               const float* __restrict__ bias, float* __restrict__ out,
                                                               int D0, int D1, int D2, int tiles1, int tiles2, int ntiles,
                                                               int act) {
  __shared__ __attribute__((aligned(16))) float lds[FH0 * 6 * FH2 * CIN];
  constexpr int FH1 = 6, FHV = FH0 * FH1 * FH2, NS = (FHV + 255) / 256, Cout = 24;
  __shared__ __attribute__((aligned(16))) float otile[4 * 64 * Cout];  // output staging: [wave][voxel 64][24]
  constexpr int NG = 27 * CIN * 6, NR = (NG + 15) / 16;
  constexpr uint32_t OOB = 0x80000000u;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vy = lane >> 4, vx = lane & 15;
  const int xbase = ((wave * FH1 + vy) * FH2 + vx) * CIN;
  float wr[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) wr[r] = wp[r * 64 + lane];
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)((int64_t)D0 * D1 * D2 * CIN * 4), 0x00020000);
  int rel[NS], ldsa[NS];
  uint32_t cmask[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int j = tid + 256 * i;
    const int hz = j / (FH1 * FH2), hy = (j / FH2) % FH1, hx = j % FH2;
    rel[i] = ((hz * D1 + hy) * D2 + hx) * CIN * 4;
    ldsa[i] = j * CIN;
    cmask[i] = j < FHV ? ((1u << hz) | (1u << (6 + hy)) | (1u << (12 + hx))) : 0xFFFFFFFFu;
  }
  float stg[NS][CIN];
  auto load_tile = [&](int t) {
    const int t2 = t % tiles2, t1 = (t / tiles2) % tiles1, t0 = t / (tiles2 * tiles1);
    const int z0 = t0 * FT0, y0 = t1 * 4, x0 = t2 * FT2;
    uint32_t bad = 0x80000000u;
#pragma unroll
    for (int h = 0; h < FH0; ++h) bad |= ((unsigned)(z0 - 1 + h) >= (unsigned)D0) ? (1u << h) : 0u;
#pragma unroll
    for (int h = 0; h < FH1; ++h) bad |= ((unsigned)(y0 - 1 + h) >= (unsigned)D1) ? (1u << (6 + h)) : 0u;
#pragma unroll
    for (int h = 0; h < FH2; ++h) bad |= ((unsigned)(x0 - 1 + h) >= (unsigned)D2) ? (1u << (12 + h)) : 0u;
    const int org = (((z0 - 1) * D1 + (y0 - 1)) * D2 + (x0 - 1)) * CIN * 4;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int vo = (cmask[i] & bad) ? (int)OOB : rel[i] + org;
      if constexpr (CIN == 2) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rin, vo, 0, 0);
        stg[i][0] = __uint_as_float(v.x);
        stg[i][1] = __uint_as_float(v.y);
      } else {
        stg[i][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rin, vo, 0, 0));
      }
    }
  };
  const int G = gridDim.x;
  const int my_pos = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  if (my_pos < ntiles) load_tile(my_pos);
  for (int t = my_pos; t < ntiles; t += G) {
    const int t2 = t % tiles2, t1 = (t / tiles2) % tiles1, t0 = t / (tiles2 * tiles1);
    const int z0 = t0 * FT0, y0 = t1 * 4, x0 = t2 * FT2;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      if (i < NS - 1 || tid + 256 * i < FHV) {
#pragma unroll
        for (int c = 0; c < CIN; ++c) lds[ldsa[i] + c] = stg[i][c];
      }
    }
    __syncthreads();
    if (t + G < ntiles) load_tile(t + G);
    f32x4 acc[6];
#pragma unroll
    for (int g = 0; g < 6; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float xq[2][CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) xq[0][c] = lds[xbase + c];
    sfor<0, 27>([&](auto T) {
      constexpr int tap = decltype(T)::value, tn = tap + 1 < 27 ? tap + 1 : tap;
      constexpr int noff = (((tn / 9) * FH1 + (tn / 3) % 3) * FH2 + tn % 3) * CIN;
#pragma unroll
      for (int c = 0; c < CIN; ++c) xq[(tap + 1) & 1][c] = lds[xbase + noff + c];
      sfor<0, CIN * 6>([&](auto GI) {
        constexpr int gi = decltype(GI)::value, ci = gi / 6, g = gi % 6, GG = (tap * CIN + ci) * 6 + g;
        acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[GG / 16], xq[tap & 1][ci], acc[g], 4, GG % 16, 0);
      });
    });
    // Epilogue through LDS (round 4): a lane holds the 24 channels of ONE voxel, so a float4 store per lane wrote 64 pieces of
    // 16 bytes 96 bytes apart (this kernel is bound by its 393 MB of output: 2.4 TB/s).  The wave's 64 voxels x 96 B are four
    // x-rows of 1536 contiguous bytes each in memory: written to the wave's own 6 KB of LDS voxel by voxel and read back as 384
    // consecutive 16-byte pieces, six per lane, every store instruction covers 1 KB of consecutive addresses.
    float* ot = otile + wave * (64 * Cout);
    // (the wave's 6 KB are its own: a wave barrier + a wavefront-scope fence order this tile's writes after the previous tile's
    // read-back and before the read-back below -- no reliance on the DS queue's in-order execution or on instruction order)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g = 0; g < 6; ++g) {
      float4 v = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
      if (bias) {
        v.x += bias[4 * g];
        v.y += bias[4 * g + 1];
        v.z += bias[4 * g + 2];
        v.w += bias[4 * g + 3];
      }
      if (act == 1) {
        v.x = elu_f(v.x);
        v.y = elu_f(v.y);
        v.z = elu_f(v.z);
        v.w = elu_f(v.w);
      }
      *reinterpret_cast<float4*>(ot + lane * Cout + 4 * g) = v;
    }
    const int gz = z0 + wave;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int q = lane + 64 * k;           // 16-byte piece of the wave's tile: row q / 96, piece c of the row
      const int r = q / 96, c = q - r * 96;
      const float4 v = *reinterpret_cast<const float4*>(ot + q * 4);
      if (gz < D0 && y0 + r < D1 && x0 + c / 6 < D2)
        *reinterpret_cast<float4*>(out + (((size_t)gz * D1 + y0 + r) * D2 + x0) * Cout + c * 4) = v;
    }
  }
}

// ---- VALU-lean forward kernel for the non-persistent cases (CK = 24, NT <= 3): deep levels (MT = 2, optional
// split-K) and the nearest-upsample-folded parity convolutions (NTAPS = 8).  Same tile, LDS layout and packed weights
// as conv3d_fwd_kernel; the address arithmetic is taken off the vector ALU as in the persistent kernel:
//  * the tile of a workgroup is fixed, so the per-thread halo columns (byte offset incl. tile origin, LDS address,
//    validity) are computed once; the channel chunk, the parity origin and the z-plane travel in the scalar soffset of
//    raw buffer loads, zero padding is the hardware range check;
//  * taps are unrolled statically (27, or the 2x2x2 window of a parity conv whose position inside the 3x3x3 stencil is a
//    per-parity shift of the LDS base and of the scalar weight offset), so LDS reads and weight loads use immediates
//    and the A/B fragments ping-pong between statically indexed register sets.
template <int NT, int MT, bool KSPLIT, int NTAPS>
__global__ __launch_bounds__(256, 2) void conv3d_fwd_lean_kernel(const float* __restrict__ in,
                                                                 const float* __restrict__ wp,
                                                                 const float* __restrict__ bias, float* __restrict__ out,
                                                                 int D0, int D1, int D2, int Cin, int Cout, int ncc,
                                                                 int tiles1, int tiles2, int act, ConvExt ext) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [FHV][CKP]
  constexpr int CK = 24, FT1 = MT, FH1 = MT + 2, CKP = CK + 4, NCG = CK / 8, C4 = CK / 4;
  constexpr int PL4 = FH1 * FH2 * C4, NJ = (PL4 + 255) / 256, NLD = NJ * FH0;
  constexpr uint32_t OOB = 0x80000000u;
  static_assert(NTAPS == 27 || !KSPLIT, "parity convs are not split over K");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int t;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, j = b >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int t2 = t % tiles2;
  t /= tiles2;
  const int t1 = t % tiles1;
  const int t0 = t / tiles1;
  const int z0 = t0 * FT0, y0 = t1 * FT1, x0 = t2 * FT2;
  const int nc = blockIdx.y;
  const int mode = (NTAPS == 8) ? ext.mode : 0;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int li = lane & 15, kq = lane >> 4;
  int a_base[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) a_base[m] = ((wave * FH1 + m) * FH2 + li) * CKP + 2 * kq;

  // input view: conv-grid voxel g -> tensor voxel g*is + io (mode 2 reads one parity sub-lattice of the 2x tensor)
  const int is = (mode == 2) ? 2 : 1;
  const int sX = is * Cin * 4, sY = is * (D2 * is) * Cin * 4, sZ = is * (D1 * is) * (D2 * is) * Cin * 4;  // bytes
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(in), 0, (int)((int64_t)D0 * D1 * D2 * is * is * is * Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, 0x7FFFFFF0, 0x00020000);
  uint32_t vrel[NJ];
  int ldsa[NJ];
  {
    uint32_t bad = 0;
#pragma unroll
    for (int h = 0; h < FH1; ++h) bad |= ((unsigned)(y0 - 1 + h) >= (unsigned)D1) ? (1u << h) : 0u;
#pragma unroll
    for (int h = 0; h < FH2; ++h) bad |= ((unsigned)(x0 - 1 + h) >= (unsigned)D2) ? (1u << (8 + h)) : 0u;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      const int j = tid + 256 * i;
      const int hy = j / (FH2 * C4), r = j - hy * (FH2 * C4), hx = r / C4, c4 = r - hx * C4;
      const bool ok = (j < PL4) && !(((1u << hy) | (1u << (8 + hx))) & bad);
      vrel[i] = ok ? (uint32_t)((y0 - 1 + hy) * sY + (x0 - 1 + hx) * sX + c4 * 16) : OOB;
      ldsa[i] = (hy * FH2 + hx) * CKP + c4 * 4;
    }
  }
  constexpr int TAPB = NCG * NT * 512;  // bytes of packed weights per tap
  auto bload = [&](int soff, int idx) -> float2 {
    const int hi = idx >> 2, lo = idx & 3;
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rw, lane * 8 + lo * 512, soff + hi * 2048, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
  };

  const int cpz = KSPLIT ? (ncc + (int)gridDim.z - 1) / (int)gridDim.z : ncc;
  const int cc_lo = KSPLIT ? (int)blockIdx.z * cpz : 0;
  const int cc_hi = min(ncc, cc_lo + cpz);
  // mode 2 sums 8 parity convs into one output: all in this workgroup, or (small deep levels: too few tiles to fill
  // the chip) split over gridDim.z workgroups that accumulate with atomics onto a zeroed output
  const bool psplit = (NTAPS == 8) && mode == 2 && gridDim.z > 1;
  const int npar = (mode == 2) ? 8 / (int)gridDim.z : 1;
  const int par_lo = (mode == 2) ? (int)blockIdx.z * npar : 0;
  const int opar = (mode == 1) ? (int)blockIdx.z : 0;
  // halo of iteration `it` (parity, chunk) -> staging registers; issued one iteration ahead so that the loads complete
  // behind the MFMAs of the previous chunk
  const int nit = npar * (cc_hi - cc_lo);
  float4 stg[NLD];
  auto halo_loads = [&](int it) {
    const int ipar = it / (cc_hi - cc_lo);
    const int cc = cc_lo + it - ipar * (cc_hi - cc_lo);
    const int par = (mode == 1) ? opar : ipar + par_lo;
    const int pconst = (mode == 2) ? ((((par >> 2) & 1) * (D1 * 2) + ((par >> 1) & 1)) * (D2 * 2) + (par & 1)) * Cin * 4 : 0;
#pragma unroll
    for (int hz = 0; hz < FH0; ++hz) {
      const int gz = z0 - 1 + hz;
      const bool pv = (unsigned)gz < (unsigned)D0;
      const int so = (pv ? gz * sZ : 0) + cc * (CK * 4) + pconst;
#pragma unroll
      for (int i = 0; i < NJ; ++i) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)(pv ? vrel[i] : OOB), so, 0);
        stg[hz * NJ + i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
    }
  };
  if (nit > 0) halo_loads(0);
  for (int it = 0; it < nit; ++it) {
    const int ipar = it / (cc_hi - cc_lo);
    const int cc = cc_lo + it - ipar * (cc_hi - cc_lo);
    const int par = (mode == 1) ? opar : ipar + par_lo;
    const int pz = (par >> 2) & 1, py = (par >> 1) & 1, px = par & 1;
    // position of the 2x2x2 window inside the 3x3x3 stencil (see up_tapmask): mode 1 shift = parity, mode 2 (flipped
    // taps) shift = 1 - parity
    const int shz = (NTAPS == 8) ? (mode == 2 ? 1 - pz : pz) : 0, shy = (NTAPS == 8) ? (mode == 2 ? 1 - py : py) : 0,
              shx = (NTAPS == 8) ? (mode == 2 ? 1 - px : px) : 0;
    const int shtap = (shz * 3 + shy) * 3 + shx;
    const int shlds = ((shz * FH1 + shy) * FH2 + shx) * CKP;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      if (i < NJ - 1 || tid + 256 * i < PL4) {
#pragma unroll
        for (int hz = 0; hz < FH0; ++hz)
          *reinterpret_cast<float4*>(&lds[ldsa[i] + hz * (FH1 * FH2 * CKP)]) = stg[hz * NJ + i];
      }
    }
    __syncthreads();
    if (it + 1 < nit) halo_loads(it + 1);

    const int wsoff = (int)(((int64_t)(nc * ncc + cc) * 27 + shtap) * TAPB + (int64_t)par * ext.wstride * 4);
    int ab[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) ab[m] = a_base[m] + shlds;
    auto tap_id = [](int i) { return NTAPS == 27 ? i : ((i >> 2) & 1) * 9 + ((i >> 1) & 1) * 3 + (i & 1); };
    auto tap_lds = [&](int i) {
      const int tt = tap_id(i);
      return (((tt / 9) * FH1 + (tt / 3) % 3) * FH2 + tt % 3) * CKP;
    };
    float2 bb[2][NCG][NT];
    float2 aa[2][MT];
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
      for (int n = 0; n < NT; ++n) bb[0][g][n] = bload(wsoff + tap_id(0) * TAPB, g * NT + n);
#pragma unroll
    for (int m = 0; m < MT; ++m) aa[0][m] = *reinterpret_cast<const float2*>(&lds[ab[m] + tap_lds(0)]);
#pragma unroll
    for (int ti = 0; ti < NTAPS; ++ti) {
      if (ti + 1 < NTAPS) {
#pragma unroll
        for (int g = 0; g < NCG; ++g)
#pragma unroll
          for (int n = 0; n < NT; ++n) bb[(ti + 1) & 1][g][n] = bload(wsoff + tap_id(ti + 1) * TAPB, g * NT + n);
      }
      const int toff = tap_lds(ti);
      const int toff_n = tap_lds(ti + 1 < NTAPS ? ti + 1 : ti);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < NCG; ++g) {
        const int st = ti * NCG + g;
        const int noff = (g + 1 < NCG) ? toff + (g + 1) * 8 : toff_n;
#pragma unroll
        for (int m = 0; m < MT; ++m) aa[(st + 1) & 1][m] = *reinterpret_cast<const float2*>(&lds[ab[m] + noff]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[st & 1][m].x, bb[ti & 1][g][n].x, acc[m][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[st & 1][m].y, bb[ti & 1][g][n].y, acc[m][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- epilogue
  const int gz = z0 + wave;
  if (gz < D0) {
    const int os = (mode == 1) ? 2 : 1;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int gy = y0 + m;
      if (gy >= D1) continue;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int co = (nc * NT + n) * 16 + li;
        if (co >= Cout) continue;
        const float bv = (!KSPLIT && bias) ? bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gx = x0 + kq * 4 + r;
          if (gx < D2) {
            const size_t oidx = (((size_t)(gz * os + ((opar >> 2) & 1)) * (D1 * os) + (gy * os + ((opar >> 1) & 1))) *
                                     (D2 * os) + (gx * os + (opar & 1))) * Cout + co;
            if (KSPLIT || psplit) {
              atomicAdd(out + oidx, acc[m][n][r]);
            } else {
              float v = acc[m][n][r] + bv;
              if (act == 2) v *= elu_dy(ext.addend[oidx]);
              else if (ext.addend) v += ext.addend[oidx];
              if (act == 1) v = elu_f(v);
              out[oidx] = v;
            }
          }
        }
      }
    }
  }
}

// ---- "brick" forward kernel for the small deep levels (40^3, 20^3, 10^3; CK = 24, plain 27-tap convs) ---------------
// The 16-voxel MFMA rows of the kernels above are 16 consecutive x; on volumes of width 40 / 20 / 10 the padding to
// 16-wide tiles wastes 17 / 37 / 48 % of the matrix work.  Here an m-tile is a 4(y) x 4(x) brick of one z-plane, a tile
// is 4 z-planes of WM bricks in y (4 x 4 x 4 voxels for WM = 1: divides 40 and 20 exactly), and the WM*WN <= 4 waves are
// arranged WM x WN: WN waves share the SAME voxels and split the output channels (each loads only its own B
// fragments), so a 64-voxel tile still carries 4 x NT x 16 output channels of work per staged halo.  The small 6x(TY+2)x6
// halo tile makes the launch fine-grained enough for split-K to fill the chip.  Everything else (buffer-load staging with
// hardware zero padding, static taps, register ping-pong, packed weights) is conv3d_fwd_lean_kernel's.
template <int NT, int WM, int WN, bool KSPLIT, int NTAPS>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv3d_fwd_brick_kernel(const float* __restrict__ in,
                                                                  const float* __restrict__ wp,
                                                                  const float* __restrict__ bias, float* __restrict__ out,
                                                                  int D0, int D1, int D2, int Cin, int Cout, int ncc,
                                                                  int tiles1, int tiles2, int act, ConvExt ext) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CK = 24, CKP = CK + 4, NCG = CK / 8, C4 = CK / 4, MT = 4;
  constexpr int NTHR = 64 * WM * WN, TZ = 4, TY = 4 * WM, TX = 4;  // tile in voxels
  constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
  constexpr int PL4 = HY * HX * C4, NJ = (PL4 + NTHR - 1) / NTHR, NLD = NJ * HZ;
  constexpr uint32_t OOB = 0x80000000u;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int t = blockIdx.x;
  const int t2 = t % tiles2;
  t /= tiles2;
  const int t1 = t % tiles1;
  const int t0 = t / tiles1;
  const int z0 = t0 * TZ, y0 = t1 * TY, x0 = t2 * TX;
  const int nc = blockIdx.y * WN + wn;  // this wave's group of NT n-tiles
  static_assert(NTAPS == 27 || !KSPLIT, "parity convs are not split over K");
  const int mode = (NTAPS == 8) ? ext.mode : 0;  // 1: up-forward (parity = blockIdx.z), 2: up-data-gradient (8 parities)
  const float* addend = ext.addend;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int li = lane & 15, kq = lane >> 4;
  // m-tile m of this wave = brick (z = m, y-block = wm): voxel li -> (dy = li >> 2, dx = li & 3)
  int a_base[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) a_base[m] = ((m * HY + wm * 4 + (li >> 2)) * HX + (li & 3)) * CKP + 2 * kq;

  const int is = (mode == 2) ? 2 : 1;  // input view: conv-grid voxel g -> tensor voxel g*is + parity
  const int sX = is * Cin * 4, sY = is * (D2 * is) * Cin * 4, sZ = is * (D1 * is) * (D2 * is) * Cin * 4;  // bytes
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(in), 0, (int)((int64_t)D0 * D1 * D2 * is * is * is * Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, 0x7FFFFFF0, 0x00020000);
  uint32_t vrel[NJ];
  int ldsa[NJ];
  {
    uint32_t bad = 0;
#pragma unroll
    for (int h = 0; h < HY; ++h) bad |= ((unsigned)(y0 - 1 + h) >= (unsigned)D1) ? (1u << h) : 0u;
#pragma unroll
    for (int h = 0; h < HX; ++h) bad |= ((unsigned)(x0 - 1 + h) >= (unsigned)D2) ? (1u << (20 + h)) : 0u;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      const int j = tid + NTHR * i;
      const int hy = j / (HX * C4), r = j - hy * (HX * C4), hx = r / C4, c4 = r - hx * C4;
      const bool ok = (j < PL4) && !(((1u << hy) | (1u << (20 + hx))) & bad);
      vrel[i] = ok ? (uint32_t)((y0 - 1 + hy) * sY + (x0 - 1 + hx) * sX + c4 * 16) : OOB;
      ldsa[i] = (hy * HX + hx) * CKP + c4 * 4;
    }
  }
  constexpr int TAPB = NCG * NT * 512;
  auto bload = [&](int soff, int idx) -> float2 {
    const int hi = idx >> 2, lo = idx & 3;
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rw, lane * 8 + lo * 512, soff + hi * 2048, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
  };
  const int cpz = KSPLIT ? (ncc + (int)gridDim.z - 1) / (int)gridDim.z : ncc;
  const int cc_lo = KSPLIT ? (int)blockIdx.z * cpz : 0;
  const int cc_hi = min(ncc, cc_lo + cpz);
  const bool psplit = (NTAPS == 8) && mode == 2 && gridDim.z > 1;  // parities over gridDim.z workgroups (atomics)
  const int npar = (mode == 2) ? 8 / (int)gridDim.z : 1;
  const int par_lo = (mode == 2) ? (int)blockIdx.z * npar : 0;
  const int opar = (mode == 1) ? (int)blockIdx.z : 0;
  const int nit = npar * (cc_hi - cc_lo);
  float4 stg[NLD];
  auto halo_loads = [&](int it) {
    const int ipar = it / (cc_hi - cc_lo) + par_lo;
    const int cc = cc_lo + (it % (cc_hi - cc_lo));
    const int pconst =
        (mode == 2) ? ((((ipar >> 2) & 1) * (D1 * 2) + ((ipar >> 1) & 1)) * (D2 * 2) + (ipar & 1)) * Cin * 4 : 0;
#pragma unroll
    for (int hz = 0; hz < HZ; ++hz) {
      const int gz = z0 - 1 + hz;
      const bool pv = (unsigned)gz < (unsigned)D0;
      const int so = (pv ? gz * sZ : 0) + cc * (CK * 4) + pconst;
#pragma unroll
      for (int i = 0; i < NJ; ++i) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)(pv ? vrel[i] : OOB), so, 0);
        stg[hz * NJ + i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
    }
  };
  if (nit > 0) halo_loads(0);
  for (int it = 0; it < nit; ++it) {
    const int ipar = it / (cc_hi - cc_lo);
    const int cc = cc_lo + it - ipar * (cc_hi - cc_lo);
    const int par = (mode == 1) ? opar : ipar + par_lo;
    const int pz = (par >> 2) & 1, py = (par >> 1) & 1, px = par & 1;
    // position of the 2x2x2 window inside the 3x3x3 stencil: mode 1 shift = parity, mode 2 (flipped taps) 1 - parity
    const int shz = (NTAPS == 8) ? (mode == 2 ? 1 - pz : pz) : 0, shy = (NTAPS == 8) ? (mode == 2 ? 1 - py : py) : 0,
              shx = (NTAPS == 8) ? (mode == 2 ? 1 - px : px) : 0;
    const int shtap = (shz * 3 + shy) * 3 + shx;
    const int shlds = ((shz * HY + shy) * HX + shx) * CKP;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      if (i < NJ - 1 || tid + NTHR * i < PL4) {
#pragma unroll
        for (int hz = 0; hz < HZ; ++hz) *reinterpret_cast<float4*>(&lds[ldsa[i] + hz * (HY * HX * CKP)]) = stg[hz * NJ + i];
      }
    }
    __syncthreads();
    if (it + 1 < nit) halo_loads(it + 1);

    const int wsoff = (int)(((int64_t)(nc * ncc + cc) * 27 + shtap) * TAPB + (int64_t)par * ext.wstride * 4);
    int ab[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) ab[m] = a_base[m] + shlds;
    auto tap_id = [](int i) { return NTAPS == 27 ? i : ((i >> 2) & 1) * 9 + ((i >> 1) & 1) * 3 + (i & 1); };
    auto tap_lds = [&](int i) {
      const int tt = tap_id(i);
      return (((tt / 9) * HY + (tt / 3) % 3) * HX + tt % 3) * CKP;
    };
    float2 bb[2][NCG][NT];
    float2 aa[2][MT];
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
      for (int n = 0; n < NT; ++n) bb[0][g][n] = bload(wsoff + tap_id(0) * TAPB, g * NT + n);
#pragma unroll
    for (int m = 0; m < MT; ++m) aa[0][m] = *reinterpret_cast<const float2*>(&lds[ab[m] + tap_lds(0)]);
#pragma unroll
    for (int ti = 0; ti < NTAPS; ++ti) {
      if (ti + 1 < NTAPS) {
#pragma unroll
        for (int g = 0; g < NCG; ++g)
#pragma unroll
          for (int n = 0; n < NT; ++n) bb[(ti + 1) & 1][g][n] = bload(wsoff + tap_id(ti + 1) * TAPB, g * NT + n);
      }
      const int toff = tap_lds(ti);
      const int toff_n = tap_lds(ti + 1 < NTAPS ? ti + 1 : ti);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < NCG; ++g) {
        const int st = ti * NCG + g;
        const int noff = (g + 1 < NCG) ? toff + (g + 1) * 8 : toff_n;
#pragma unroll
        for (int m = 0; m < MT; ++m) aa[(st + 1) & 1][m] = *reinterpret_cast<const float2*>(&lds[ab[m] + noff]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[st & 1][m].x, bb[ti & 1][g][n].x, acc[m][n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[st & 1][m].y, bb[ti & 1][g][n].y, acc[m][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- epilogue: D row = kq*4 + r -> brick voxel (dy = row >> 2, dx = row & 3), col = li -> channel
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int gz = z0 + m;
    if (gz >= D0) continue;
    const int gy = y0 + wm * 4 + kq;  // row >> 2 = kq
    if (gy >= D1) continue;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int co = (nc * NT + n) * 16 + li;
      if (co >= Cout) continue;
      const float bv = (!KSPLIT && bias) ? bias[co] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gx = x0 + r;  // row & 3 = r
        if (gx < D2) {
          const int os = (mode == 1) ? 2 : 1;
          const size_t oidx = (((size_t)(gz * os + ((opar >> 2) & 1)) * (D1 * os) + (gy * os + ((opar >> 1) & 1))) *
                                   (D2 * os) + (gx * os + (opar & 1))) * Cout + co;
          if (KSPLIT || psplit) {
            atomicAdd(out + oidx, acc[m][n][r]);
          } else {
            float v = acc[m][n][r] + bv;
            if (act == 2) v *= elu_dy(addend[oidx]);
            else if (addend) v += addend[oidx];
            if (act == 1) v = elu_f(v);
            out[oidx] = v;
          }
        }
      }
    }
  }
}

// bias + activation after a split-K accumulation
__global__ __launch_bounds__(256) void bias_act_kernel(float* __restrict__ y, const float* __restrict__ bias, int64_t n,
                                                       int C, int act, const float* __restrict__ eluy) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = y[i] + (bias ? bias[i % C] : 0.f);
    if (act == 1) v = elu_f(v);
    if (act == 2) v *= elu_dy(eluy[i]);
    y[i] = v;
  }
}

// -------------------------------------------------------------------------------------------- weight gradient
constexpr int WT0 = 2, WT1 = 4, WT2 = 16;  // voxel tile = 128 voxels = 32 k-steps
constexpr int WH0 = 4, WH1 = 6, WH2 = 18;
constexpr int WHV = WH0 * WH1 * WH2;        // 432
constexpr int WVPX = 434;                   // 434/2 = 217 odd -> rows of [ci][voxel] land on distinct even banks
constexpr int WTV = WT0 * WT1 * WT2;        // 128
constexpr int WVPD = 130;                   // 65 odd

struct WgExt {
  int up;          // 1: nearest-upsample folding — parity = blockIdx.y % 8, dout is read on that parity sub-lattice of
                   // the 2x tensor, the 8 active taps of the parity are the GEMM rows, dw += parity * dwstride
  int cin_total;   // row length of dw in input channels
  int ci_off;      // first input channel of this layer part inside dw
  int64_t dwstride;
  int dbg;         // timing experiments: 8 = skip the flush
  float* dbias;    // optional: += sum over voxels of dout (a constant-1 GEMM row), else nullptr
  int64_t det_stride;  // deterministic mode: dw / dbias point at per-workgroup-column planes, plane blockIdx.x = + det_stride
                       // floats (0 otherwise: every workgroup adds into the one dw); see det_prepare / det_finish
};

template <int CK, int NT, int MS, int NTAPS>
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_kernel(const float* __restrict__ in,
                                                              const float* __restrict__ dout, float* __restrict__ dw,
                                                              int D0, int D1, int D2, int Cin, int Cout, int tiles0,
                                                              int tiles1, int tiles2, WgExt ext) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lx = lds;                 // [CK][WVPX]
  float* ld = lds + CK * WVPX;     // [NT*16][WVPD]
  constexpr int MR = NTAPS * CK;   // GEMM rows (tap, ci)
  constexpr int MTILES = (MR + 15) / 16;
  constexpr int MTP = (MTILES + MS - 1) / MS;  // m-tiles per workgroup (the GEMM rows are split over MS workgroups)
  constexpr int MTW = (MTP + 3) / 4;           // m-tiles per wave
  constexpr int C4 = CK / 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  static_assert(NTAPS == 27 || MS == 1, "the parity variant does not split the GEMM rows");
  const int par = (NTAPS == 8) ? (int)(blockIdx.y & 7) : 0;
  const int cc = (NTAPS == 8) ? (int)(blockIdx.y >> 3) : (int)(blockIdx.y / MS);  // input-channel chunk
  const int mt0 = (NTAPS == 8) ? 0 : (int)(blockIdx.y % MS) * MTP;  // first m-tile of this workgroup
  const uint32_t tapmask = (NTAPS == 8) ? up_tapmask(par, false) : 0x7FFFFFFu;
  auto nth_tap = [&](int i) {  // i-th active tap of the mask
    if (NTAPS == 27) return i;
    uint32_t m = tapmask;
    for (int k = 0; k < i; ++k) m &= m - 1;
    return (int)__builtin_ctz(m);
  };
  const int ds = (NTAPS == 8) ? 2 : 1;  // dout view: low-res grid voxel g -> tensor voxel 2g + parity
  const int dp0 = (par >> 2) & 1, dp1 = (par >> 1) & 1, dp2 = par & 1;
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
    int r = (mt0 + wave + 4 * m) * 16 + li;
    if (r >= MR) r = MR - 1;
    const int ti = r / CK, cil = r - ti * CK;
    const int tap = nth_tap(ti);
    const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
    a_base[m] = cil * WVPX + (dz * WH1 + dy) * WH2 + dx;
  }
  const int b_base = li * WVPD + kq;

  const int ntiles = tiles0 * tiles1 * tiles2;
  constexpr int NX = (WHV * C4 + 255) / 256;
  constexpr int ND = (WTV * NT * 4 + 255) / 256;
  // Global loads of one tile (x halo + dy), branch-free (clamped address + select).  The slot arithmetic goes
  // through an opaque copy of tid so that hipcc recomputes it per call instead of hoisting 14 x 4 registers.
  auto load_x = [&](int k, int z0, int y0, int x0) -> float4 {
    int tv = tid;
    asm volatile("" : "+v"(tv));
    const int f = tv + k * 256;
    const int vox = f / C4, c4 = f - vox * C4;
    const int hx = vox % WH2, hy = (vox / WH2) % WH1, hz = vox / (WH2 * WH1);
    const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
    const int c = cc * CK + c4 * 4;
    const bool ok = (f < WHV * C4) & (gz >= 0) & (gz < D0) & (gy >= 0) & (gy < D1) & (gx >= 0) & (gx < D2);
    const size_t off = ok ? ((((size_t)gz * D1 + gy) * D2 + gx) * Cin + c) : 0;
    float4 v;
    if constexpr (CK == 24) {
      v = ld4(in + off);
      if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const float* src = in + off;
      v.x = (ok && c + 0 < Cin) ? src[0] : 0.f;
      v.y = (ok && c + 1 < Cin) ? src[(c + 1 < Cin) ? 1 : 0] : 0.f;
      v.z = (ok && c + 2 < Cin) ? src[(c + 2 < Cin) ? 2 : 0] : 0.f;
      v.w = (ok && c + 3 < Cin) ? src[(c + 3 < Cin) ? 3 : 0] : 0.f;
    }
    return v;
  };
  auto load_d = [&](int k, int z0, int y0, int x0) -> float4 {
    int tv = tid;
    asm volatile("" : "+v"(tv));
    const int f = tv + k * 256;
    const int vox = f / (NT * 4), c4 = f - vox * (NT * 4);
    const int vx = vox % WT2, vy = (vox / WT2) % WT1, vz = vox / (WT2 * WT1);
    const int gz = z0 + vz, gy = y0 + vy, gx = x0 + vx;
    const int c = co0 + c4 * 4;
    const bool ok = (f < WTV * NT * 4) & (gz < D0) & (gy < D1) & (gx < D2);
    const size_t off =
        ok ? ((((size_t)(gz * ds + dp0) * (D1 * ds) + (gy * ds + dp1)) * (D2 * ds) + (gx * ds + dp2)) * Cout + c) : 0;
    float4 v;
    if (vec_out) {
      const bool okc = ok && c < Cout;
      v = ld4(dout + (okc ? off : 0));
      if (!okc) v = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const float* src = dout + off;
      v.x = (ok && c + 0 < Cout) ? src[0] : 0.f;
      v.y = (ok && c + 1 < Cout) ? src[(c + 1 < Cout) ? 1 : 0] : 0.f;
      v.z = (ok && c + 2 < Cout) ? src[(c + 2 < Cout) ? 2 : 0] : 0.f;
      v.w = (ok && c + 3 < Cout) ? src[(c + 3 < Cout) ? 3 : 0] : 0.f;
    }
    return v;
  };
  auto tile_origin = [&](int t, int& z0, int& y0, int& x0) {
    const int t2 = t % tiles2, t1 = (t / tiles2) % tiles1, t0 = t / (tiles2 * tiles1);
    z0 = t0 * WT0;
    y0 = t1 * WT1;
    x0 = t2 * WT2;
  };
  float4 sx[NX], sd[ND];
  if ((int)blockIdx.x < ntiles) {
    int z0, y0, x0;
    tile_origin(blockIdx.x, z0, y0, x0);
#pragma unroll
    for (int k = 0; k < NX; ++k) sx[k] = load_x(k, z0, y0, x0);
#pragma unroll
    for (int k = 0; k < ND; ++k) sd[k] = load_d(k, z0, y0, x0);
  }
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    __syncthreads();  // previous tile fully consumed
    // ---- registers -> LDS, transposed: x [ci][voxel], dy [co][voxel]
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      int tv = tid;
      asm volatile("" : "+v"(tv));
      const int f = tv + k * 256;
      const int vox = f / C4, c4 = f - vox * C4;
      if (f < WHV * C4) {
        float* d = lx + (c4 * 4) * WVPX + vox;
        d[0] = sx[k].x;
        d[WVPX] = sx[k].y;
        d[2 * WVPX] = sx[k].z;
        d[3 * WVPX] = sx[k].w;
      }
    }
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      int tv = tid;
      asm volatile("" : "+v"(tv));
      const int f = tv + k * 256;
      const int vox = f / (NT * 4), c4 = f - vox * (NT * 4);
      if (f < WTV * NT * 4) {
        float* d = ld + (c4 * 4) * WVPD + vox;
        d[0] = sd[k].x;
        d[WVPD] = sd[k].y;
        d[2 * WVPD] = sd[k].z;
        d[3 * WVPD] = sd[k].w;
      }
    }
    __syncthreads();
    // ---- prefetch the next tile of this workgroup into the (now free) staging registers; the loads complete
    // while the 32 k-steps below keep the matrix cores busy
    if (t + (int)gridDim.x < ntiles) {
      int z0, y0, x0;
      tile_origin(t + gridDim.x, z0, y0, x0);
#pragma unroll
      for (int k = 0; k < NX; ++k) sx[k] = load_x(k, z0, y0, x0);
#pragma unroll
      for (int k = 0; k < ND; ++k) sd[k] = load_d(k, z0, y0, x0);
    }
    // ---- 32 k-steps of 4 voxels; the LDS reads of step ks+1 are issued before the MFMAs of step ks
    {
      float acur[MTW], anext[MTW], bcur[NT], bnext[NT];
      {
        const int voff = (((kq >> 6) * WH1 + ((kq >> 4) & 3)) * WH2) + (kq & 15);
#pragma unroll
        for (int n = 0; n < NT; ++n) bcur[n] = ld[b_base + n * 16 * WVPD];
#pragma unroll
        for (int m = 0; m < MTW; ++m) acur[m] = lx[a_base[m] + voff];
      }
      for (int ks = 0; ks < WTV / 4; ++ks) {
        const int kn = (ks + 1 < WTV / 4) ? ks + 1 : ks;
        const int k = kn * 4 + kq;
        const int voff = (((k >> 6) * WH1 + ((k >> 4) & 3)) * WH2) + (k & 15);
#pragma unroll
        for (int n = 0; n < NT; ++n) bnext[n] = ld[b_base + n * 16 * WVPD + kn * 4];
#pragma unroll
        for (int m = 0; m < MTW; ++m) anext[m] = lx[a_base[m] + voff];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[m], bcur[n], acc[m][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MTW; ++m) acur[m] = anext[m];
#pragma unroll
        for (int n = 0; n < NT; ++n) bcur[n] = bnext[n];
      }
    }
  }

  // ---- flush: D row = (lane>>4)*4 + reg -> (tap, ci), col = lane&15 -> co.  Every address is touched by ONE lane of the
  // workgroup; deterministic mode gives each workgroup column (blockIdx.x) its own plane (det_stride), summed in order later
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (wave + 4 * m >= MTP) continue;  // m-tile belongs to the next workgroup of the split
      const int row = (mt0 + wave + 4 * m) * 16 + kq * 4 + r;
      if (row >= MR) continue;
      const int ti = row / CK, cil = row - ti * CK;
      const int tap = nth_tap(ti);
      const int ci = cc * CK + cil;
      if (ci >= Cin) continue;
      float* dwp = dw + (size_t)par * ext.dwstride + (size_t)blockIdx.x * ext.det_stride;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int co = co0 + n * 16 + li;
        if (co < Cout) atomicAdd(&dwp[((size_t)tap * ext.cin_total + ext.ci_off + ci) * Cout + co], acc[m][n][r]);
      }
    }
  }
}

// ---- VALU-lean weight gradient (CK = 24, Cout % 4 == 0, tensors < 2 GiB).  Same tiling and LDS layout as
// conv3d_wgrad_kernel; what changes is everything next to the MFMAs (a vector-ALU instruction costs 3-4 cycles of
// MFMA issue on gfx950, tools/ubench/mfma_issue.hip):
//  * per-thread staging columns (global byte offset, LDS address, validity bit) are computed once per kernel; per tile
//    only "offset or out-of-range" selects remain, z-planes / row pairs travel in the scalar soffset of raw buffer
//    loads, zero padding comes from the hardware range check;
//  * the 32 k-steps are fully unrolled: LDS reads use immediate offsets and the A/B fragments ping-pong between two
//    statically indexed register sets (no copies).
template <int NT, int MS, int NTAPS>
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_lean_kernel(const float* __restrict__ in,
                                                                   const float* __restrict__ dout,
                                                                   float* __restrict__ dw, int D0, int D1, int D2,
                                                                   int Cin, int Cout, int tiles0, int tiles1, int tiles2,
                                                                   WgExt ext) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CK = 24;
  float* lx = lds;                    // [CK + 1][WVPX]; row CK is all ones (the dbias GEMM row)
  float* ld = lds + (CK + 1) * WVPX;  // [NT*16][WVPD]
  for (int i = threadIdx.x; i < WVPX; i += 256) lx[CK * WVPX + i] = 1.f;
  constexpr int MR = NTAPS * CK;
  constexpr int MTILES = (MR + 15) / 16;
  constexpr int MTP = (MTILES + MS - 1) / MS;
  constexpr int MTW = (MTP + 3) / 4;
  constexpr int C4 = CK / 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  static_assert(NTAPS == 27 || MS == 1, "the parity variant does not split the GEMM rows");
  const int par = (NTAPS == 8) ? (int)(blockIdx.y & 7) : 0;
  const int cc = (NTAPS == 8) ? (int)(blockIdx.y >> 3) : (int)(blockIdx.y / MS);
  const int mt0 = (NTAPS == 8) ? 0 : (int)(blockIdx.y % MS) * MTP;
  const uint32_t tapmask = (NTAPS == 8) ? up_tapmask(par, false) : 0x7FFFFFFu;
  auto nth_tap = [&](int i) {
    if (NTAPS == 27) return i;
    uint32_t m = tapmask;
    for (int k = 0; k < i; ++k) m &= m - 1;
    return (int)__builtin_ctz(m);
  };
  const int ds = (NTAPS == 8) ? 2 : 1;
  const int dp0 = (par >> 2) & 1, dp1 = (par >> 1) & 1, dp2 = par & 1;
  const int co0 = blockIdx.z * NT * 16;
  constexpr uint32_t OOB = 0x80000000u;

  f32x4 acc[MTW][NT];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // A rows r = (mt0 + wave + 4m)*16 + li -> (tap, ci); the k index of the lane (kq) is folded into the base
  int a_base[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    int r = (mt0 + wave + 4 * m) * 16 + li;
    const bool ones = NTAPS == 27 && r == MR;  // row MR: x = 1 at the centre tap -> sum of dout = dbias
    if (r >= MR) r = MR - 1;
    const int ti = r / CK, cil = r - ti * CK;
    const int tap = ones ? 13 : nth_tap(ti);
    a_base[m] = (ones ? CK : cil) * WVPX + ((tap / 9) * WH1 + (tap / 3) % 3) * WH2 + tap % 3 + kq;
  }
  const int b_base = li * WVPD + kq;

  // ---- x halo: 4 z-planes of 6 x 18 voxels x 6 channel quads = 648 float4 per plane, 3 columns per thread
  constexpr int PL4 = WH1 * WH2 * C4, NJ = (PL4 + 255) / 256, NX = NJ * WH0;
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)((int64_t)D0 * D1 * D2 * Cin * 4), 0x00020000);
  int xrel[NJ], xlds[NJ];
  uint32_t xmask[NJ];
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    const int j = tid + 256 * i;
    const int hy = j / (WH2 * C4), r = j - hy * (WH2 * C4), hx = r / C4, c4 = r - hx * C4;
    xrel[i] = ((hy * D2 + hx) * Cin + c4 * 4) * 4;
    xlds[i] = (c4 * 4) * WVPX + hy * WH2 + hx;
    xmask[i] = j < PL4 ? ((1u << hy) | (1u << (8 + hx))) : 0xFFFFFFFFu;
  }
  const int xplane = D1 * D2 * Cin * 4;
  // ---- dy tile: 8 (z,y) rows of 16 voxels x QN channel quads; row pieces are contiguous in memory
  constexpr int QN = NT * 4, RQ = 16 * QN, ND = QN / 2;
  constexpr int P = (QN == 12) ? 3 : 1;      // period of the (row, column) pattern in i
  constexpr int RPP = 256 * P / RQ;          // rows advanced per period (4, 2, 4)
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(dout), 0, (int)((int64_t)D0 * D1 * D2 * ds * ds * ds * Cout * 4), 0x00020000);
  const int dsx = ds * Cout * 4, dsy = ds * (D2 * ds) * Cout * 4;  // byte strides of the dy view
  const int64_t dsz = (int64_t)ds * (D1 * ds) * (D2 * ds) * Cout * 4;
  const int dconst = ((dp0 * (D1 * ds) + dp1) * (D2 * ds) + dp2) * Cout * 4;
  int drel[P], dlds[P];
  uint32_t dmask[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int j = tid + 256 * p;
    const int row = j / RQ, w = j - row * RQ, vx = w / QN, c4 = w - vx * QN;
    drel[p] = row * dsy + vx * dsx + (co0 + c4 * 4) * 4;   // row < 4: vz = 0, vy = row
    dlds[p] = (c4 * 4) * WVPD + row * WT2 + vx;
    dmask[p] = (co0 + c4 * 4 < Cout) ? ((1u << row) | (1u << (8 + vx))) : 0xFFFFFFFFu;
  }

  const int ntiles = tiles0 * tiles1 * tiles2;
  float4 sx[NX], sd[ND];
  auto as_f4 = [](u32x4 v) { return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)); };
  auto load_tile = [&](int t) {
    const int t2 = t % tiles2, t1 = (t / tiles2) % tiles1, t0 = t / (tiles2 * tiles1);
    const int z0 = t0 * WT0, y0 = t1 * WT1, x0 = t2 * WT2;
    // x halo
    uint32_t bad = 0x80000000u;
#pragma unroll
    for (int h = 0; h < WH1; ++h) bad |= ((unsigned)(y0 - 1 + h) >= (unsigned)D1) ? (1u << h) : 0u;
#pragma unroll
    for (int h = 0; h < WH2; ++h) bad |= ((unsigned)(x0 - 1 + h) >= (unsigned)D2) ? (1u << (8 + h)) : 0u;
    const int yx = (((y0 - 1) * D2 + (x0 - 1)) * Cin + cc * CK) * 4;
    uint32_t xo[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) xo[i] = (xmask[i] & bad) ? OOB : (uint32_t)(xrel[i] + yx);
#pragma unroll
    for (int hz = 0; hz < WH0; ++hz) {
      const int gz = z0 - 1 + hz;
      const bool pv = (unsigned)gz < (unsigned)D0;
      const int so = pv ? gz * xplane : 0;
#pragma unroll
      for (int i = 0; i < NJ; ++i)
        sx[hz * NJ + i] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rin, (int)(pv ? xo[i] : OOB), so, 0));
    }
    // dy rows
    uint32_t dbad = 0x80000000u;
#pragma unroll
    for (int h = 0; h < WT1; ++h) dbad |= (y0 + h >= D1) ? (1u << h) : 0u;
#pragma unroll
    for (int h = 0; h < WT2; ++h) dbad |= (x0 + h >= D2) ? (1u << (8 + h)) : 0u;
    const int64_t dbase = (int64_t)z0 * dsz + (int64_t)y0 * dsy + (int64_t)x0 * dsx + dconst;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int p = i % P, q = i / P;
      const int rowadd = q * RPP;            // rows 0..7: vz = rowadd >> 2 (+0), vy += rowadd & 3
      const int vz = rowadd >> 2, dvy = rowadd & 3;
      const bool pv = z0 + vz < D0;
      // validity of vy + dvy: shift the y bits of the bad mask down by dvy
      const uint32_t badq = ((dbad & 0xFFu) >> dvy) | (dbad & 0xFFFFFF00u);
      const uint32_t vo = ((dmask[p] & badq) || !pv) ? OOB : (uint32_t)drel[p];
      const int so = pv ? (int)(dbase + vz * dsz + (int64_t)dvy * dsy) : 0;
      sd[i] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rdo, (int)vo, so, 0));
    }
  };
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);

  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      if (i < NJ - 1 || tid + 256 * i < PL4) {
#pragma unroll
        for (int hz = 0; hz < WH0; ++hz) {
          float* d = lx + xlds[i] + hz * (WH1 * WH2);
          const float4 v = sx[hz * NJ + i];
          d[0] = v.x;
          d[WVPX] = v.y;
          d[2 * WVPX] = v.z;
          d[3 * WVPX] = v.w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int p = i % P, q = i / P;
      float* d = ld + dlds[p] + q * RPP * WT2;
      const float4 v = sd[i];
      d[0] = v.x;
      d[WVPD] = v.y;
      d[2 * WVPD] = v.z;
      d[3 * WVPD] = v.w;
    }
    __syncthreads();
    if (t + (int)gridDim.x < ntiles) load_tile(t + gridDim.x);  // lands while the 32 k-steps below run
    float aa[2][MTW], bb[2][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) bb[0][n] = ld[b_base + n * 16 * WVPD];
#pragma unroll
    for (int m = 0; m < MTW; ++m) aa[0][m] = lx[a_base[m]];
#pragma unroll
    for (int ks = 0; ks < WTV / 4; ++ks) {
      constexpr int LAST = WTV / 4 - 1;
      const int kn = ks < LAST ? ks + 1 : LAST;
      const int voff = ((kn >> 4) * WH1 + ((kn >> 2) & 3)) * WH2 + 4 * (kn & 3);
#pragma unroll
      for (int n = 0; n < NT; ++n) bb[(ks + 1) & 1][n] = ld[b_base + n * 16 * WVPD + kn * 4];
#pragma unroll
      for (int m = 0; m < MTW; ++m) aa[(ks + 1) & 1][m] = lx[a_base[m] + voff];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[ks & 1][m], bb[ks & 1][n], acc[m][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- flush: D row = (lane>>4)*4 + reg -> (tap, ci), col = lane&15 -> co
  if (ext.dbg & 8) return;
  const size_t detoff = (size_t)blockIdx.x * ext.det_stride;  // deterministic mode: this workgroup column's own plane
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (wave + 4 * m >= MTP) continue;
      const int row = (mt0 + wave + 4 * m) * 16 + kq * 4 + r;
      if (NTAPS == 27 && row == MR && ext.dbias && cc == 0) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int co = co0 + n * 16 + li;
          if (co < Cout) atomicAdd(&ext.dbias[detoff + co], acc[m][n][r]);
        }
      }
      if (row >= MR) continue;
      const int ti = row / CK, cil = row - ti * CK;
      const int tap = nth_tap(ti);
      const int ci = cc * CK + cil;
      if (ci >= Cin) continue;
      float* dwp = dw + (size_t)par * ext.dwstride + detoff;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int co = co0 + n * 16 + li;
        if (co < Cout) atomicAdd(&dwp[((size_t)tap * ext.cin_total + ext.ci_off + ci) * Cout + co], acc[m][n][r]);
      }
    }
  }
}

// ---- weight gradient with small box tiles for the deep levels (40^3: 4x4x8 voxels, 20^3: 4x4x4) ---------------------
// conv3d_wgrad_lean_kernel's 2x4x16 tiles pad x = 40 / 20 to 48 / 32 (17 / 37 % of the k-steps wasted on zeros).  Same
// GEMM (rows = (tap, ci) + the constant-1 dbias row, N = co, K = voxels), LDS layouts and unrolled k-steps; the tile is
// TZ x TY x TX with TX a multiple of 4 so that the 4 voxels of a k-step share a (z, y) row, and the dy tile is staged
// through generic per-thread slots (voxel-major) instead of the row-pattern of the 16-wide tile.
template <int NT, int MS, int TZ, int TY, int TX>
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_box_kernel(const float* __restrict__ in,
                                                                  const float* __restrict__ dout, float* __restrict__ dw,
                                                                  int D0, int D1, int D2, int Cin, int Cout, int tiles0,
                                                                  int tiles1, int tiles2, WgExt ext) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CK = 24, C4 = CK / 4;
  constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HV = HZ * HY * HX, TV = TZ * TY * TX;
  constexpr int VPX = HV + ((HV / 2) % 2 == 0 ? 2 : 0) + (HV % 2);  // even, half of it odd: conflict-free ds_read_b32
  constexpr int VPD = TV + 2;
  static_assert((VPX / 2) % 2 == 1 && (VPD / 2) % 2 == 1 && TX % 4 == 0, "LDS strides / k-step rows");
  float* lx = lds;                   // [CK + 1][VPX]; row CK = ones (dbias row)
  float* ld = lds + (CK + 1) * VPX;  // [NT*16][VPD]
  for (int i = threadIdx.x; i < VPX; i += 256) lx[CK * VPX + i] = 1.f;
  constexpr int MR = 27 * CK, MTILES = (MR + 15) / 16 + ((MR % 16) == 0 ? 1 : 0);
  constexpr int MTP = (MTILES + MS - 1) / MS, MTW = (MTP + 3) / 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int cc = (int)(blockIdx.y / MS);
  const int mt0 = (int)(blockIdx.y % MS) * MTP;
  const int co0 = blockIdx.z * NT * 16;
  constexpr uint32_t OOB = 0x80000000u;

  f32x4 acc[MTW][NT];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int a_base[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    int r = (mt0 + wave + 4 * m) * 16 + li;
    const bool ones = r == MR;
    if (r >= MR) r = MR - 1;
    const int tap = ones ? 13 : r / CK, cil = r - (r / CK) * CK;
    a_base[m] = (ones ? CK : cil) * VPX + ((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3 + kq;
  }
  const int b_base = li * VPD + kq;

  // ---- x halo: HZ planes of HY x HX voxels x 6 quads
  constexpr int PL4 = HY * HX * C4, NJ = (PL4 + 255) / 256, NX = NJ * HZ;
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)((int64_t)D0 * D1 * D2 * Cin * 4), 0x00020000);
  int xrel[NJ], xlds[NJ];
  uint32_t xmask[NJ];
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    const int j = tid + 256 * i;
    const int hy = j / (HX * C4), r = j - hy * (HX * C4), hx = r / C4, c4 = r - hx * C4;
    xrel[i] = ((hy * D2 + hx) * Cin + c4 * 4) * 4;
    xlds[i] = (c4 * 4) * VPX + hy * HX + hx;
    xmask[i] = j < PL4 ? ((1u << hy) | (1u << (12 + hx))) : 0xFFFFFFFFu;
  }
  const int xplane = D1 * D2 * Cin * 4;
  // ---- dy tile: TV voxels x QN quads, slot j = tid + 256 i -> (voxel = j / QN, quad = j % QN)
  constexpr int QN = NT * 4, ND = (TV * QN + 255) / 256;
  const __amdgpu_buffer_rsrc_t rdo =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dout), 0, (int)((int64_t)D0 * D1 * D2 * Cout * 4), 0x00020000);
  int drel[ND], dlds[ND];
  uint32_t dmask[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int j = tid + 256 * i;
    const int vox = j / QN, c4 = j - vox * QN;
    const int vx = vox % TX, vy = (vox / TX) % TY, vz = vox / (TX * TY);
    drel[i] = (((vz * D1 + vy) * D2 + vx) * Cout + co0 + c4 * 4) * 4;
    dlds[i] = (c4 * 4) * VPD + vox;
    dmask[i] = (j < TV * QN && co0 + c4 * 4 < Cout) ? ((1u << vz) | (1u << (8 + vy)) | (1u << (16 + vx))) : 0xFFFFFFFFu;
  }

  const int ntiles = tiles0 * tiles1 * tiles2;
  float4 sx[NX], sd[ND];
  auto as_f4 = [](u32x4 v) { return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)); };
  auto load_tile = [&](int t) {
    const int t2 = t % tiles2, t1 = (t / tiles2) % tiles1, t0 = t / (tiles2 * tiles1);
    const int z0 = t0 * TZ, y0 = t1 * TY, x0 = t2 * TX;
    uint32_t bad = 0x80000000u;
#pragma unroll
    for (int h = 0; h < HY; ++h) bad |= ((unsigned)(y0 - 1 + h) >= (unsigned)D1) ? (1u << h) : 0u;
#pragma unroll
    for (int h = 0; h < HX; ++h) bad |= ((unsigned)(x0 - 1 + h) >= (unsigned)D2) ? (1u << (12 + h)) : 0u;
    const int yx = (((y0 - 1) * D2 + (x0 - 1)) * Cin + cc * CK) * 4;
    uint32_t xo[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) xo[i] = (xmask[i] & bad) ? OOB : (uint32_t)(xrel[i] + yx);
#pragma unroll
    for (int hz = 0; hz < HZ; ++hz) {
      const int gz = z0 - 1 + hz;
      const bool pv = (unsigned)gz < (unsigned)D0;
      const int so = pv ? gz * xplane : 0;
#pragma unroll
      for (int i = 0; i < NJ; ++i)
        sx[hz * NJ + i] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rin, (int)(pv ? xo[i] : OOB), so, 0));
    }
    uint32_t dbad = 0x80000000u;
#pragma unroll
    for (int h = 0; h < TZ; ++h) dbad |= (z0 + h >= D0) ? (1u << h) : 0u;
#pragma unroll
    for (int h = 0; h < TY; ++h) dbad |= (y0 + h >= D1) ? (1u << (8 + h)) : 0u;
#pragma unroll
    for (int h = 0; h < TX; ++h) dbad |= (x0 + h >= D2) ? (1u << (16 + h)) : 0u;
    const int dbase = (((z0 * D1 + y0) * D2 + x0) * Cout) * 4;
#pragma unroll
    for (int i = 0; i < ND; ++i)
      sd[i] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rdo, (int)((dmask[i] & dbad) ? OOB : (uint32_t)drel[i]), dbase, 0));
  };
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);

  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      if (i < NJ - 1 || tid + 256 * i < PL4) {
#pragma unroll
        for (int hz = 0; hz < HZ; ++hz) {
          float* d = lx + xlds[i] + hz * (HY * HX);
          const float4 v = sx[hz * NJ + i];
          d[0] = v.x;
          d[VPX] = v.y;
          d[2 * VPX] = v.z;
          d[3 * VPX] = v.w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      if (i < ND - 1 || tid + 256 * i < TV * QN) {
        float* d = ld + dlds[i];
        const float4 v = sd[i];
        d[0] = v.x;
        d[VPD] = v.y;
        d[2 * VPD] = v.z;
        d[3 * VPD] = v.w;
      }
    }
    __syncthreads();
    if (t + (int)gridDim.x < ntiles) load_tile(t + gridDim.x);
    float aa[2][MTW], bb[2][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) bb[0][n] = ld[b_base + n * 16 * VPD];
#pragma unroll
    for (int m = 0; m < MTW; ++m) aa[0][m] = lx[a_base[m]];
#pragma unroll
    for (int ks = 0; ks < TV / 4; ++ks) {
      constexpr int LAST = TV / 4 - 1;
      const int kn = ks < LAST ? ks + 1 : LAST;
      const int k0 = kn * 4;  // first voxel of the next k-step: (z, y) row and x offset
      const int voff = ((k0 / (TX * TY)) * HY + (k0 / TX) % TY) * HX + k0 % TX;
#pragma unroll
      for (int n = 0; n < NT; ++n) bb[(ks + 1) & 1][n] = ld[b_base + n * 16 * VPD + kn * 4];
#pragma unroll
      for (int m = 0; m < MTW; ++m) aa[(ks + 1) & 1][m] = lx[a_base[m] + voff];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[ks & 1][m], bb[ks & 1][n], acc[m][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (ext.dbg & 8) return;
  const size_t detoff = (size_t)blockIdx.x * ext.det_stride;  // deterministic mode: this workgroup column's own plane
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (wave + 4 * m >= MTP) continue;
      const int row = (mt0 + wave + 4 * m) * 16 + kq * 4 + r;
      if (row == MR && ext.dbias && cc == 0) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int co = co0 + n * 16 + li;
          if (co < Cout) atomicAdd(&ext.dbias[detoff + co], acc[m][n][r]);
        }
      }
      if (row >= MR) continue;
      const int tap = row / CK, cil = row - tap * CK;
      const int ci = cc * CK + cil;
      if (ci >= Cin) continue;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int co = co0 + n * 16 + li;
        if (co < Cout) atomicAdd(&dw[detoff + ((size_t)tap * ext.cin_total + ext.ci_off + ci) * Cout + co], acc[m][n][r]);
      }
    }
  }
}

// ---- first-layer weight gradient (Cin <= 2, Cout = 24) on the 4x4x1 MFMA -------------------------------------------
// dW[(tap,ci), co] = sum_v x[v+tap, ci] dz[v, co]: 27*Cin <= 54 GEMM rows.  A (broadcast via ABID) = dz: 8 consecutive
// voxels x 24 channels are 192 contiguous floats = exactly three coalesced registers (register r, lane l <-> float
// 64r + l = 4G + i with group G = voxel*6 + channel-quad), no LDS needed.  B = x: lane = row (tap, ci) reads its own
// element of the voxel's neighbourhood from the [648][Cin] LDS halo tile (per-lane row offset + immediate voxel
// offset).  One MFMA per (voxel, channel quad); D[i] of lane r = dW[row r][4g+i], kept in 24 registers per lane over
// all tiles of the workgroup.  Lane 27*Cin reads a constant 1 and so accumulates dbias = sum_v dz[v].
template <int CIN>
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_c2_kernel(const float* __restrict__ in,
                                                                 const float* __restrict__ dout,
                                                                 float* __restrict__ partial, int D0, int D1, int D2,
                                                                 int tiles1, int tiles2, int ntiles) {
  constexpr int FH1 = 6, FHV = FH0 * FH1 * FH2, NS = (FHV + 255) / 256, Cout = 24, NROW = 27 * CIN;
  __shared__ __attribute__((aligned(16))) float lds[2 * FHV * CIN];  // halo tile | ones (read by the dbias lane)
  __shared__ float red[3 * 64 * 24];
  constexpr uint32_t OOB = 0x80000000u;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < FHV * CIN; i += 256) lds[FHV * CIN + i] = 1.f;
  // per-lane B address: row (tap, ci) of this lane inside the wave's z-plane of the halo tile
  int rbase;
  {
    const int r = lane < NROW ? lane : 0;
    const int tap = r / CIN, ci = r - tap * CIN;
    rbase = (((wave + tap / 9) * FH1 + (tap / 3) % 3) * FH2 + tap % 3) * CIN + ci;
    if (lane == NROW) rbase = FHV * CIN + wave * FH1 * FH2 * CIN;  // ones
  }
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)((int64_t)D0 * D1 * D2 * CIN * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dout), 0, (int)((int64_t)D0 * D1 * D2 * Cout * 4), 0x00020000);
  int rel[NS], ldsa[NS];
  uint32_t cmask[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int j = tid + 256 * i;
    const int hz = j / (FH1 * FH2), hy = (j / FH2) % FH1, hx = j % FH2;
    rel[i] = ((hz * D1 + hy) * D2 + hx) * CIN * 4;
    ldsa[i] = j * CIN;
    cmask[i] = j < FHV ? ((1u << hz) | (1u << (6 + hy)) | (1u << (12 + hx))) : 0xFFFFFFFFu;
  }
  int kk[3];  // voxel (0..7) of the octet that register r of this lane belongs to
#pragma unroll
  for (int r = 0; r < 3; ++r) kk[r] = (64 * r + lane) / 24;
  float stg[NS][CIN];
  float an[8][3];  // dz of the next tile (this wave's z-plane): 8 octets x 3 registers, requested one tile ahead
  auto load_tile = [&](int t) {
    const int t2 = t % tiles2, t1 = (t / tiles2) % tiles1, t0 = t / (tiles2 * tiles1);
    const int z0 = t0 * FT0, y0 = t1 * 4, x0 = t2 * FT2;
    {
      const int gz = z0 + wave;
#pragma unroll
      for (int o = 0; o < 8; ++o) {  // octet o: row vy = o >> 1, x half o & 1
        const int gy = y0 + (o >> 1), xs = x0 + 8 * (o & 1);
        const bool rowok = gz < D0 && gy < D1;  // scalar
        const int so = rowok ? (((gz * D1 + gy) * D2 + xs) * Cout) * 4 : 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int vo = (rowok && xs + kk[r] < D2) ? (64 * r + lane) * 4 : (int)OOB;
          an[o][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rdo, vo, so, 0));
        }
      }
    }
    uint32_t bad = 0x80000000u;
#pragma unroll
    for (int h = 0; h < FH0; ++h) bad |= ((unsigned)(z0 - 1 + h) >= (unsigned)D0) ? (1u << h) : 0u;
#pragma unroll
    for (int h = 0; h < FH1; ++h) bad |= ((unsigned)(y0 - 1 + h) >= (unsigned)D1) ? (1u << (6 + h)) : 0u;
#pragma unroll
    for (int h = 0; h < FH2; ++h) bad |= ((unsigned)(x0 - 1 + h) >= (unsigned)D2) ? (1u << (12 + h)) : 0u;
    const int org = (((z0 - 1) * D1 + (y0 - 1)) * D2 + (x0 - 1)) * CIN * 4;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int vo = (cmask[i] & bad) ? (int)OOB : rel[i] + org;
      if constexpr (CIN == 2) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rin, vo, 0, 0);
        stg[i][0] = __uint_as_float(v.x);
        stg[i][1] = __uint_as_float(v.y);
      } else {
        stg[i][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rin, vo, 0, 0));
      }
    }
  };
  f32x4 acc[6];
#pragma unroll
  for (int g = 0; g < 6; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int G = gridDim.x;
  const int my_pos = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  if (my_pos < ntiles) load_tile(my_pos);
  for (int t = my_pos; t < ntiles; t += G) {
    const int t2 = t % tiles2, t1 = (t / tiles2) % tiles1, t0 = t / (tiles2 * tiles1);
    const int z0 = t0 * FT0, y0 = t1 * 4, x0 = t2 * FT2;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      if (i < NS - 1 || tid + 256 * i < FHV) {
#pragma unroll
        for (int c = 0; c < CIN; ++c) lds[ldsa[i] + c] = stg[i][c];
      }
    }
    __syncthreads();
    float ar[8][3];
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
      for (int r = 0; r < 3; ++r) ar[o][r] = an[o][r];
    if (t + G < ntiles) load_tile(t + G);
    sfor<0, 8>([&](auto O) {
      constexpr int o = decltype(O)::value;
      float xv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) xv[k] = lds[rbase + (((o >> 1) * FH2) + 8 * (o & 1) + k) * CIN];
      sfor<0, 48>([&](auto GI) {
        constexpr int GG = decltype(GI)::value, k = GG / 6, g = GG % 6;
        acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(ar[o][GG / 16], xv[k], acc[g], 4, GG % 16, 0);
      });
    });
  }
  // ---- combine the four waves (z-planes) and flush once per workgroup
  __syncthreads();
  if (wave > 0) {
#pragma unroll
    for (int g = 0; g < 6; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[((wave - 1) * 24 + g * 4 + i) * 64 + lane] = acc[g][i];
  }
  __syncthreads();
  if (wave == 0) {  // partial[workgroup][lane = GEMM row][24]; rows_reduce_kernel sums the workgroups
    float* dst = partial + ((size_t)blockIdx.x * 64 + lane) * 24;
#pragma unroll
    for (int g = 0; g < 6; ++g) {
      float4 v;
      float* pv = &v.x;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = g * 4 + i;
        pv[i] = acc[g][i] + red[c * 64 + lane] + red[(24 + c) * 64 + lane] + red[(48 + c) * 64 + lane];
      }
      *reinterpret_cast<float4*>(dst + 4 * g) = v;
    }
  }
}

// second stage of the first-layer weight gradient: dw / dbias += sum over workgroups of partial[wg][row][24].
// (Thousands of workgroups adding to the same 1320 addresses with atomics serialise on the memory-side atomic units;
// here every address receives gridDim.y adds.)
__global__ void rows_reduce_kernel(const float* __restrict__ partial, int nwg, float* __restrict__ dw,
                                   float* __restrict__ dbias, int nrow, int cin, int cin_total, int ci_off) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;  // row * 24 + c
  const bool live = j < (nrow + 1) * 24;
  const int row = j / 24, c = j - row * 24;
  const int per = (nwg + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nwg, b0 + per);
  float v = 0.f;
  if (live)
    for (int b = b0; b < b1; ++b) v += partial[(size_t)b * 1536 + j];
  int* turn = syn_turn_begin(blockIdx.x, blockIdx.y);  // chain = address block (x), position = workgroup group (y)
  if (live) {
    if (row < nrow) {
      const int tap = row / cin, ci = row - tap * cin;
      atomicAdd(&dw[((size_t)tap * cin_total + ci_off + ci) * 24 + c], v);
    } else if (dbias) {
      atomicAdd(&dbias[c], v);
    }
  }
  syn_turn_end(turn, blockIdx.y, gridDim.y);
}

// ---- weight gradient of the forward parity convs (folded decoder conv, Cout = 24) on the 4x4x1 MFMA ----------------
// conv3d_wgrad_p4_kernel for the parity convs of a folded decoder conv: the 2x2x2 window x 6 channel quads of ONE output
// parity is exactly 48 blocks = 3 B registers (no idle block slots).  Each of the 4 waves owns one parity (blockIdx.y picks
// the parity half) and walks all 256 low-res voxels of the staged tile, so one halo tile feeds 4 x 256 x 18 MFMAs -- with
// the waves splitting the voxels of a single parity instead (first version) the same tile fed a quarter of that and the
// matrix pipes were busy 59 % of the time.  dz is read on the wave's parity sub-lattice of the hi-res tensor (voxel
// stride 2), which only changes the per-lane offsets of the three coalesced A registers.  grid = (gx, 2 x chunks).
__global__ __launch_bounds__(256, 2) void conv3d_up_wgrad_p4_kernel(const float* __restrict__ in,
                                                                    const float* __restrict__ dout,
                                                                    float* __restrict__ dwc, int D0, int D1, int D2,
                                                                    int Cin, int tiles1, int tiles2, int ntiles,
                                                                    int64_t dwstride, int dbg, int64_t det_stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CK = 24, MT = 4, Cout = 24;
  constexpr int FT1 = MT, FH1 = MT + 2, CKP = CK + 4, C4 = CK / 4, NQ = 3;
  constexpr uint32_t OOB = 0x80000000u;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int par = (blockIdx.y & 1) * 4 + wave, cc = blockIdx.y >> 1;
  const int pz = (par >> 2) & 1, py = (par >> 1) & 1, px = par & 1;
  const int G = gridDim.x;
  const int my_pos = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);

  // B rows of this lane: block = 16 q + (lane >> 2) = window tap * 6 + quad
  int rowoff[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int blk = 16 * q + (lane >> 2);
    const int ti = blk / 6, quad = blk - ti * 6;
    const int tz = pz + ((ti >> 2) & 1), ty = py + ((ti >> 1) & 1), tx = px + (ti & 1);
    rowoff[q] = ((tz * FH1 + ty) * FH2 + tx) * CKP + quad * 4 + (lane & 3);
  }
  constexpr int PLANE4 = FH1 * FH2 * C4, NJ = (PLANE4 + 255) / 256, NLD = NJ * FH0;
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)((int64_t)D0 * D1 * D2 * Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(dout), 0, (int)((int64_t)D0 * D1 * D2 * 8 * Cout * 4), 0x00020000);
  int rel[NJ], ldsa[NJ];
  uint32_t cmask[NJ];
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    const int j = tid + 256 * i;
    const int hy = j / (FH2 * C4), r = j - hy * (FH2 * C4), hx = r / C4, c4 = r - hx * C4;
    rel[i] = ((hy * D2 + hx) * Cin + c4 * 4) * 4;
    ldsa[i] = ((hy * FH2 + hx) * CKP + c4 * 4);
    cmask[i] = j < PLANE4 ? ((1u << hy) | (1u << (8 + hx))) : 0xFFFFFFFFu;
  }
  const int plane_bytes = D1 * D2 * Cin * 4;
  int kk[3], aoff[3];  // dz register r, lane l <-> float 64 r + l = 24 k + c of an octet; voxel stride 2 in the hi-res row
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int f = 64 * r + lane;
    kk[r] = f / 24;
    aoff[r] = (kk[r] * 2 * Cout + (f - kk[r] * 24)) * 4;
  }
  float4 stg[NLD];
  auto tile_origin = [&](int t, int& z0, int& y0, int& x0) {
    const int t2 = t % tiles2, t1 = (t / tiles2) % tiles1, t0 = t / (tiles2 * tiles1);
    z0 = t0 * FT0;
    y0 = t1 * FT1;
    x0 = t2 * FT2;
  };
  auto load_halo = [&](int t) {
    int z0, y0, x0;
    tile_origin(t, z0, y0, x0);
    uint32_t bad = 0x80000000u;
#pragma unroll
    for (int h = 0; h < FH1; ++h) bad |= ((unsigned)(y0 - 1 + h) >= (unsigned)D1) ? (1u << h) : 0u;
#pragma unroll
    for (int h = 0; h < FH2; ++h) bad |= ((unsigned)(x0 - 1 + h) >= (unsigned)D2) ? (1u << (8 + h)) : 0u;
    const int yx = (((y0 - 1) * D2 + (x0 - 1)) * Cin + cc * CK) * 4;
    uint32_t voff[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) voff[i] = (cmask[i] & bad) ? OOB : (uint32_t)(rel[i] + yx);
#pragma unroll
    for (int hz = 0; hz < FH0; ++hz) {
      const int gz = z0 - 1 + hz;
      const bool pv = (unsigned)gz < (unsigned)D0;
#pragma unroll
      for (int i = 0; i < NJ; ++i) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)(pv ? voff[i] : OOB), pv ? gz * plane_bytes : 0, 0);
        stg[hz * NJ + i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
    }
  };
  // dz of z-plane `z` of tile (z0, y0, x0) on this wave's parity: 8 octets (row y = o >> 1, x half o & 1) x 3 registers
  auto load_dz = [&](int z0, int y0, int x0, int z, float (&a)[8][3]) {
    const int gz = z0 + z;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      const int gy = y0 + (o >> 1), xs = x0 + 8 * (o & 1);
      const bool ok = gz < D0 && gy < D1;  // scalar
      const int so = ok ? ((((2 * gz + pz) * (2 * D1) + (2 * gy + py)) * (2 * D2) + (2 * xs + px)) * Cout) * 4 : 0;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int vo = (ok && xs + kk[r] < D2) ? aoff[r] : (int)OOB;
        a[o][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rdo, vo, so, 0));
      }
    }
  };

  f32x4 acc[NQ][6];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int g = 0; g < 6; ++g) acc[q][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float an[8][3];
  int z0 = 0, y0 = 0, x0 = 0;
  if (my_pos < ntiles) {
    load_halo(my_pos);
    tile_origin(my_pos, z0, y0, x0);
    load_dz(z0, y0, x0, 0, an);
  }
  for (int t = my_pos; t < ntiles; t += G) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      if (i < NJ - 1 || tid + 256 * i < PLANE4) {
#pragma unroll
        for (int hz = 0; hz < FH0; ++hz)
          *reinterpret_cast<float4*>(&lds[ldsa[i] + hz * (FH1 * FH2 * CKP)]) = stg[hz * NJ + i];
      }
    }
    __syncthreads();
    const bool has_next = t + G < ntiles;
    if (has_next) load_halo(t + G);
    int nz0 = z0, ny0 = y0, nx0 = x0;
    if (has_next) tile_origin(t + G, nz0, ny0, nx0);
    for (int z = 0; z < FT0; ++z) {
      float ar[8][3];
#pragma unroll
      for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int r = 0; r < 3; ++r) ar[o][r] = an[o][r];
      if (z + 1 < FT0) {
        load_dz(z0, y0, x0, z + 1, an);
      } else if (has_next) {
        load_dz(nz0, ny0, nx0, 0, an);
      }
      int rb[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) rb[q] = rowoff[q] + z * (FH1 * FH2 * CKP);
      float xq[2][NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) xq[0][q] = lds[rb[q]];
      sfor<0, 64>([&](auto S) {
        constexpr int sv = decltype(S)::value, o = sv / 8, k = sv % 8;
        constexpr int sn = sv + 1 < 64 ? sv + 1 : sv, on = sn / 8, kn = sn % 8;
        constexpr int nbase = ((on >> 1) * FH2 + 8 * (on & 1) + kn) * CKP;
#pragma unroll
        for (int q = 0; q < NQ; ++q) xq[(sv + 1) & 1][q] = lds[rb[q] + nbase];
        __builtin_amdgcn_sched_barrier(0);
        sfor<0, 6>([&](auto GI) {
          constexpr int g = decltype(GI)::value, GG = k * 6 + g;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            acc[q][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(ar[o][GG / 16], xq[sv & 1][q], acc[q][g], 4, GG % 16, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    z0 = nz0;
    y0 = ny0;
    x0 = nx0;
  }
  if (dbg & 8) return;
  // ---- flush, one wave (= one parity) at a time through LDS: its [8 taps][24 ci][24 co] partial is laid out linearly so
  // that every global atomic instruction covers 64 consecutive floats
  dwc += (size_t)blockIdx.x * det_stride;  // deterministic mode: this workgroup column's own plane (0 otherwise)
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int blk = 16 * q + (lane >> 2);
        float* d = lds + (blk * 4 + (lane & 3)) * Cout;  // row = ti*24 + quad*4 + j
#pragma unroll
        for (int g = 0; g < 6; ++g)
          *reinterpret_cast<float4*>(d + 4 * g) = make_float4(acc[q][g][0], acc[q][g][1], acc[q][g][2], acc[q][g][3]);
      }
    }
    __syncthreads();
    const int wp = (blockIdx.y & 1) * 4 + w, wz = (wp >> 2) & 1, wy = (wp >> 1) & 1, wx = wp & 1;
    float* dst = dwc + (size_t)wp * dwstride;
    for (int e = tid; e < 8 * CK * Cout; e += 256) {
      const int ti = e / (CK * Cout), r = e - ti * (CK * Cout);
      const int tap = ((wz + ((ti >> 2) & 1)) * 3 + (wy + ((ti >> 1) & 1))) * 3 + (wx + (ti & 1));
      atomicAdd(dst + ((size_t)tap * Cin + cc * CK) * Cout + r, lds[e]);
    }
  }
}

// dbias fallback for the generic weight-gradient kernel: per-channel sum of dout [n][C]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t n, int C,
                                                     float* __restrict__ out) {
  // threads run along the channels (coalesced rows of C floats), 256 / C voxel rows in flight per workgroup
  __shared__ float part[256];
  const int rows = C <= 256 ? 256 / C : 1;
  int* turn = nullptr;
  for (int c0 = 0; c0 < C; c0 += 256) {  // one pass unless C > 256
    const int width = min(256, C - c0);
    const int r = threadIdx.x / width, c = threadIdx.x - r * width;
    float acc = 0.f;
    if (r < rows)
      for (int64_t v = (int64_t)blockIdx.x * rows + r; v < n; v += (int64_t)gridDim.x * rows) acc += x[v * C + c0 + c];
    part[threadIdx.x] = acc;
    __syncthreads();
    if (c0 == 0) turn = syn_turn_begin(0, blockIdx.x);  // deterministic mode: the whole flush of this workgroup is one turn
    if (threadIdx.x < width) {
      float t = 0.f;
      for (int k = 0; k < rows; ++k) t += part[k * width + threadIdx.x];
      atomicAdd(out + c0 + threadIdx.x, t);
    }
    __syncthreads();
  }
  syn_turn_end(turn, blockIdx.x, gridDim.x);
}

// ---- per-device library state (a process may drive several devices; every lookup is by the CURRENT device) ----------------
// Nothing here changes what a call computes: scratch buffers grown on demand, and the deterministic mode's planes / tickets
// (synthsr_hip_tuning.h: synthsr_set_deterministic is per device).
constexpr int SYN_MAX_DEVICES = 64;
struct SynDeviceState {
  int det = 0;                    // synthsr_set_deterministic: ordered sums, no split-K / parity-split forward
  SynDet* det_state = nullptr;    // device block (tickets, timeout flag, scratch of the ordered reductions)
  float* det_planes = nullptr;    // private dW planes of the ordered weight-gradient flush: CALLER memory
  size_t det_planes_floats = 0;   //   (synthsr_set_deterministic_workspace), never allocated here
  unsigned long long det_demand_bytes = 0;  // largest plane demand a weight gradient of this device has asked for
};
static SynDeviceState g_dev[SYN_MAX_DEVICES];
static SynDeviceState g_dev_invalid;   // device ids >= SYN_MAX_DEVICES / a failed hipGetDevice: never deterministic, no planes
static SynDeviceState& dev_state() {
  int d = -1;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= SYN_MAX_DEVICES) return g_dev_invalid;
  return g_dev[d];
}
static inline int det_on() { return dev_state().det; }

// ---- plan parameters.  Rounds 1-4 had a process-wide A/B switch behind each of them (synthsr_conv3d_set_option); every one was
// measured against its alternative (docs/DESIGN_NOTES_r01_r02.md, profiles/r03_*, r04_*) and the library now ships the winners
// as constants -- a plan is a pure function of (shape, channels, kind, the context's arithmetic, the device's deterministic mode).
constexpr int PLAN_PERSIST = 1;            // persistent forward kernel on the large levels
constexpr int PLAN_FORCE_MT = 0;           // (was a diagnostic override of the tile height)
constexpr int PLAN_HYBRID = 0;             // MFMA + VALU co-execution for Cout % 16 == 8: correct but slower than padding (hipcc
                                           // serialises the scalar weight loads behind the MFMAs); kept for the record, off
constexpr int PLAN_DBG = 0;                // (was the ablation mask)
constexpr int PLAN_KS_TARGET = 1024;       // workgroup target of the split-K heuristic
constexpr int PLAN_BRICK = 1;              // brick tiles (4x4 voxels per MFMA row block) on the small deep levels
constexpr int PLAN_P4 = 1;                 // 4x4x1-MFMA kernels for the Cout = 24 layers (no padding to 32 columns)
constexpr int PLAN_SPLIT_MIN_WGS = 200;    // smallest layer (4x4x16 tiles x co-chunks) on the split kernels (20^3 x 192: -25 %)
constexpr int PLAN_SPLIT_WGRAD_MIN_TILES = 1;  // split weight gradient at every size (20^3 / 10^3: 4-29 % faster than fp32 MFMA)
constexpr int PLAN_STACK24 = 1;            // stacked weight layout of the plain Cout = 24 split convs (10 instead of 12 MFMAs)
constexpr int PLAN_PSPLIT = 1;             // parity split of small up-conv data gradients

// ---- the call's context (include/synthsr_hip.h: synthsr_conv_ctx).  Every extern "C" conv entry point opens a CtxScope on the
// pointer it was handed; the planners below read cfg().  The scope is thread-local and ends with the call: two threads, or two
// calls with different contexts, never see each other's arithmetic.
struct ConvCfg {
  int arith;   // 0 fp32_mfma, 1 split, 2 split9
  int split;   // arith != 0: fp32 convs through 3 x bf16 operand pieces on the bf16 matrix cores where the layer has enough tiles
  int nprod;   // 6 | 9 partial products per multiplication
  void* ws;    // the caller's scratch (synthsr_conv_ctx.workspace) and its size; the library owns no device memory
  size_t ws_bytes;
};
static thread_local ConvCfg t_cfg = {1, 1, 6, nullptr, 0};
static inline const ConvCfg& cfg() { return t_cfg; }
struct CtxScope {
  ConvCfg prev;
  bool ok;
  explicit CtxScope(const synthsr_conv_ctx* ctx) : prev(t_cfg), ok(true) {
    const int a = ctx ? ctx->arithmetic : SYNTHSR_ARITH_SPLIT;
    // reserved fields must be zero (they can then be given a meaning later); a workspace needs its size
    if (a < 0 || a > 2 || (ctx && (ctx->reserved0 || ctx->reserved[0] || ctx->reserved[1] || (ctx->workspace_bytes && !ctx->workspace)))) {
      ok = false;
      return;
    }
    t_cfg = ConvCfg{a, a ? 1 : 0, a == 2 ? 9 : 6, ctx ? ctx->workspace : nullptr, ctx ? (size_t)ctx->workspace_bytes : 0};
  }
  ~CtxScope() { t_cfg = prev; }
};
// scratch of the call in flight: a piece of the CALLER's workspace (include/synthsr_hip.h: synthsr_conv_ctx.workspace, at most
// synthsr_conv_workspace_bytes() are ever asked for).  nullptr = the context carries none / too little: SYNTHSR_EWORKSPACE.
// Calls that share a context are ordered on one stream by contract, so successive launches may reuse the same bytes.
constexpr size_t SYN_WORKSPACE_BOUND = (size_t)16 << 20;
static float* ctx_scratch(size_t bytes) {
  static_assert((size_t)2048 * 1536 * sizeof(float) <= SYN_WORKSPACE_BOUND, "first-layer weight-gradient partials");
  return (bytes <= SYN_WORKSPACE_BOUND && bytes <= cfg().ws_bytes) ? static_cast<float*>(cfg().ws) : nullptr;
}

extern "C" int syn_split_wgrad(const float* in, const float* dout, float* dw, float* dbias, const int s[3], int cin_total,
                               int ci_off, int Cin, int Cout, int nprod, hipStream_t st);
extern "C" int syn_split_upwgrad(const float* lo, const float* dout, float* dwc, const int s[3], int Cin, int Cout, int nprod,
                                 hipStream_t st);
extern "C" int syn_split_upfwd(const float* lo, const float* wp, const float* bias, const float* addend, float* out,
                               const int s[3], int Cin, int Cout, int mt, int act, int nprod, hipStream_t st);
extern "C" int syn_split_fwd_halves(const int s[3], int Cin, int nchunks, int stacked, int nprod);
extern "C" int syn_split_fwd(const float* in, const float* wp, const float* bias, const float* addend, float* out,
                             const int s[3], int Cin, int Cout, int mt, int nchunks, int act, float* stats, float* partial,
                             int upm, int stacked, int nprod, hipStream_t st);

// does the weight gradient of a plain 3x3x3 conv take the split kernel (conv_split.hip: syn_split_wgrad)?  One place: the
// dispatcher and the query synthsr_conv3d_wgrad_runs_split (what the benchmarks price a layer against) both ask here
inline bool wgrad_takes_split(const int s[3], int Cin, int Cout) {
  const int64_t vox = (int64_t)s[0] * s[1] * s[2];
  return cfg().split && (int64_t)cdiv(s[0], 4) * cdiv(s[1], 4) * cdiv(s[2], 16) >= PLAN_SPLIT_WGRAD_MIN_TILES && (Cin % 8) == 0 && (Cout % 24) == 0 &&
         vox * Cin * 4 < (1ll << 31) && vox * Cout * 4 < (1ll << 31);
}

// ... and the weight gradient of the up-sampled channel range of a folded decoder conv (conv_split.hip: syn_split_upwgrad: six
// products, 16-channel input chunks, 24-column chunks)?  Asked by the dispatcher and by synthsr_conv3d_up_wgrad_runs_split
inline bool up_wgrad_takes_split(const int s[3], int Cl, int Cout) {
  const int64_t vox = (int64_t)s[0] * s[1] * s[2];
  return cfg().split && cfg().nprod == 6 && (Cl % 16) == 0 && (Cout % 24) == 0 && vox * Cl * 4 < (1ll << 31) &&
         8 * vox * Cout * 4 < (1ll << 31);
}

struct FwdPlan {
  int nt, mt, ksplit, nchunks, ncc, ck, persist, nv, p4, c2, brick, wn, wm, split, stacked;
  // NT = 0 selects the 4x4x1 weight layout in pack_value, NT = -Cin the first-layer layout, NT = -100 - MT the split layout,
  // NT = -200 - MT / -300 - MT the split layout of a folded conv's parity set (data gradient / forward window);
  // split: 0 no, 1 plain conv, 2 folded data gradient, 3 folded forward
  int pack_nt() const { return stacked ? -400 : split == 3 ? -300 - mt : (split == 2 ? -200 - mt : (split ? -100 - mt : (c2 ? -c2 : (p4 ? 0 : nt)))); }
  int64_t mfma_count() const {
    if (split) return nchunks;  // what pack_value needs to decode the split layout
    return (p4 || c2) ? 0 : (int64_t)nchunks * ncc * 27 * (ck / 8) * nt * 128;
  }
  int64_t count() const {
    if (stacked) return (int64_t)ncc * 7 * 5 * 64 * 4;  // [cc][step 7][tile 5][lane 64] x 8 bf16
    if (split) return (int64_t)3 * nchunks * ncc * (split >= 2 ? 2 : 7) * mt * 64 * 4;  // floats (= pairs of bf16)
    if (c2) return (int64_t)((27 * c2 * 6 + 15) / 16) * 64;
    return p4 ? (int64_t)ncc * 27 * 9 * 64 : mfma_count() + (int64_t)ncc * 27 * ck * nv;
  }
};

// Launch geometry for one layer: enough workgroups to fill 256 CUs x 2 even on the deep, small levels.
// kind: 1 plain conv, 0 parity convs of the folded decoder conv (data gradient / unspecified), 2 their forward pass
inline FwdPlan plan_fwd(const int s[3], int Cin, int Cout, int kind = 1) {
  const bool plain = kind == 1;
  FwdPlan p;
  p.split = 0;
  p.stacked = 0;
  if (cfg().split && (Cin % 8) == 0 && (Cout % 8) == 0) {
    // fp32 through three bf16 pieces per operand on the bf16 matrix cores (conv_split.hip): layers with enough 4x4x16 tiles
    const int64_t vox = (int64_t)s[0] * s[1] * s[2];
    const int ntiles = cdiv(Cout, 16);
    const int mt = syn_split_plan_mt(cdiv(s[0], 4) * cdiv(s[1], 4) * cdiv(s[2], 16), ntiles, kind != 2);
    const int nchunks = cdiv(ntiles, mt);
    const int64_t wgs = (int64_t)cdiv(s[0], 4) * cdiv(s[1], 4) * cdiv(s[2], 16) * nchunks;
    // (kind 0 = data gradient of a folded decoder conv: s is the low-resolution grid, the input lives on the 2x grid)
    // (kind 2 = their forward pass: the OUTPUT lives on the 2x grid, one co-chunk of <= 48 channels)
    if (wgs >= PLAN_SPLIT_MIN_WGS && (kind == 0 ? 8 : 1) * vox * Cin * 4 < (1ll << 31) && (kind == 2 ? 8 : 1) * vox * Cout * 4 < (1ll << 31) &&
        (kind != 2 || nchunks == 1)) {
      p.split = plain ? 1 : (kind == 0 ? 2 : 3);
      p.ck = 8;
      p.ncc = Cin / 8;
      p.mt = mt;
      p.nt = mt;
      p.nchunks = nchunks;
      p.ksplit = 1;
      // the plain Cout = 24 convs (160^3: forward and data gradient): weight pieces stacked along M (conv_split.hip, STK)
      p.stacked = (plain && Cout == 24 && cfg().arith == 1 && PLAN_STACK24) ? 1 : 0;
      p.nv = p.persist = p.p4 = p.c2 = p.brick = 0;
      p.wn = p.wm = 1;
      return p;
    }
  }
  p.ck = ck_for_fwd(Cin);
  p.ncc = cdiv(Cin, p.ck);
  const int ntiles = cdiv(Cout, 16);
  auto wgs = [&](int mt, int nt) { return (int64_t)cdiv(s[0], FT0) * cdiv(s[1], mt) * cdiv(s[2], FT2) * cdiv(ntiles, nt); };
  p.mt = 4;
  int max_nt = MAX_NT;
  if (wgs(4, std::min(MAX_NT, ntiles)) < 768 || (PLAN_FORCE_MT == 2 && ntiles <= 3) || p.ck == 32) {  // ck 32: 62 KB halo tile
    p.mt = 2;
    max_nt = 3;
  }
  p.nchunks = cdiv(ntiles, max_nt);
  p.nt = cdiv(ntiles, p.nchunks);
  p.ksplit = 1;
  p.nv = 0;
  // MFMA + VALU co-execution: the matrix and vector pipes of a SIMD run concurrently, so instead of padding
  // Cout = 16 a + 8 to 16 (a + 1) MFMA columns (25 % waste at Cout = 24) the last 8 output channels are computed with
  // v_fma (weights from SGPRs, activations from the same LDS tile) in the shadow of the MFMAs of the first 16 a.
  if (PLAN_HYBRID && p.ck == 24 && p.mt == 4 && (Cout % 16) == 8 && Cout >= 24 && ntiles <= 5) {
    p.nv = 8;
    p.nchunks = 1;
    p.nt = (Cout - 8) / 16;
  }
  const bool lt2g = (int64_t)s[0] * s[1] * s[2] * Cin * 4 < (1ll << 31);  // raw buffer addressing (32-bit offsets)
  p.persist = (p.mt == 4 && p.ck == 24 && p.nt <= 3 && (Cout % 4) == 0 && PLAN_PERSIST && p.nv == 0 && lt2g) ? 1 : 0;
  // 4x4x1 layouts: plain Cout = 24 layers, and the forward parity convs of a folded decoder conv with Cout = 24
  p.p4 = (p.persist && (kind == 1 || kind == 2) && Cout == 24 && (Cin % 24) == 0 && PLAN_P4) ? 1 : 0;
  p.c2 = (plain && Cout == 24 && Cin <= 2 && lt2g && PLAN_P4) ? Cin : 0;  // first layer: 4x4x1 MFMA over K = 27*Cin
  p.brick = 0;
  p.wn = 1;
  p.wm = 1;
  if (PLAN_BRICK && p.ck == 24 && p.mt == 2 && lt2g && (Cout % 16) == 0 && p.nv == 0 && (s[0] % 4) == 0 &&
      (s[1] % 4) == 0 && (s[2] % 4) == 0 && (s[2] % 16) != 0) {
    // output channels: NT n-tiles per wave, WN waves side by side; `nchunks` (= groups of NT n-tiles) is the packing unit
    p.brick = 1;
    p.nt = (ntiles % 3 == 0) ? 3 : ((ntiles % 2 == 0) ? 2 : 1);
    p.nchunks = ntiles / p.nt;
    p.wn = (p.nchunks % 4 == 0) ? 4 : ((p.nchunks % 2 == 0) ? 2 : 1);
    p.wm = (p.wn <= 2 && (s[1] % 8) == 0) ? 2 : 1;  // bricks in y per tile: must divide the volume
    if (p.wn == 1) {  // <= 48 output channels: too little work per staged halo, the 16-wide tiles win (measured)
      p.brick = 0;
      p.wm = 1;
      p.nchunks = cdiv(ntiles, max_nt);
      p.nt = cdiv(ntiles, p.nchunks);
    }
  }
  if (p.brick) {
    const int64_t w = (int64_t)(s[0] / 4) * (s[1] / (4 * p.wm)) * (s[2] / 4) * (p.nchunks / p.wn);
    p.ksplit = 1;
    if (w < 400 && p.ncc >= 2 && plain) {
      int ks = (int)cdiv(PLAN_KS_TARGET, (int)w);
      if (ks > p.ncc) ks = p.ncc;
      if (ks > 16) ks = 16;
      if (ks >= 2 && !det_on()) p.ksplit = ks;
    }
    return p;
  }
  const int64_t w = wgs(p.mt, p.nt);
  if (w < 512 && p.ncc >= 4 && plain && p.nv == 0) {
    int ks = (int)cdiv(PLAN_KS_TARGET, (int)w);
    if (ks > p.ncc / 2) ks = p.ncc / 2;
    if (ks > 8) ks = 8;
    if (ks >= 2 && !det_on()) p.ksplit = ks;
  }
  return p;
}

// mode 2 (data gradient of a folded decoder conv = sum of 8 parity convs): number of workgroups along z the parities are
// spread over.  1 = all eight inside one workgroup (plain stores); more only when the launch would not fill the chip
// (20^3 / 10^3 levels: 120 workgroups ran at 16-34 % of the MFMA peak), then every workgroup adds its share atomically.
inline int parity_split(int64_t workgroups, const float* bias, int act, const ConvExt& ext) {
  if (!PLAN_PSPLIT || det_on() || bias != nullptr || act != 0 || ext.addend != nullptr) return 1;
  int ps = 1;
  while (ps < 8 && workgroups * ps < 400) ps *= 2;
  return ps;
}

template <int CK, int NT, int MT, bool KS, int NV = 0>
int launch_fwd(const float* in, const float* wp, const float* bias, float* out, const int s[3], int Cin, int Cout,
               const FwdPlan& pl, int act, hipStream_t st, const ConvExt& ext) {
  const int tiles0 = cdiv(s[0], FT0), tiles1 = cdiv(s[1], MT), tiles2 = cdiv(s[2], FT2);
  const size_t smem = (size_t)FH0 * (MT + 2) * FH2 * (CK + 4) * sizeof(float);
  if constexpr (CK == 24 && NT <= 3 && NV == 0) {
    const int64_t in_bytes = (int64_t)s[0] * s[1] * s[2] * (ext.mode == 2 ? 8 : 1) * Cin * 4;
    const int64_t w_bytes = pl.count() * 4 * (ext.mode ? 8 : 1);
    if (in_bytes < (1ll << 31) && w_bytes < (1ll << 31) && !(PLAN_DBG & 32)) {
      const int64_t nout = (int64_t)s[0] * s[1] * s[2] * Cout;
      if (KS) {  // split-K accumulates with atomics: onto zeros, or onto the addend when it already sits in `out`
        if (act != 2 && ext.addend && ext.addend != out) return SYNTHSR_EINVAL;
        if ((act == 2 || !ext.addend) && hipMemsetAsync(out, 0, (size_t)nout * sizeof(float), st) != hipSuccess)
          return SYNTHSR_ELAUNCH;
      }
      int gz = KS ? pl.ksplit : (ext.mode == 1 ? 8 : 1);
      if (!KS && ext.mode == 2) {  // few tiles (deep levels): the 8 parity convs of the data gradient go to separate
        gz = parity_split((int64_t)tiles0 * tiles1 * tiles2 * pl.nchunks, bias, act, ext);  // workgroups (atomics)
        if (gz > 1 && hipMemsetAsync(out, 0, (size_t)nout * sizeof(float), st) != hipSuccess) return SYNTHSR_ELAUNCH;
      }
      const dim3 grid(tiles0 * tiles1 * tiles2, pl.nchunks, gz);
      if (ext.mode == 0) {
        static SynOncePerDevice done27;
        auto k27 = conv3d_fwd_lean_kernel<NT, MT, KS, 27>;
        if (auto once_ = done27.first()) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k27), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        }
        hipLaunchKernelGGL(k27, grid, dim3(256), smem, st, in, wp, bias, out, s[0], s[1], s[2], Cin, Cout, pl.ncc, tiles1,
                           tiles2, act, ext);
      } else {
        if constexpr (!KS) {
          static SynOncePerDevice done8;
          auto k8 = conv3d_fwd_lean_kernel<NT, MT, false, 8>;
          if (auto once_ = done8.first()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k8), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
          }
          hipLaunchKernelGGL(k8, grid, dim3(256), smem, st, in, wp, bias, out, s[0], s[1], s[2], Cin, Cout, pl.ncc, tiles1,
                             tiles2, act, ext);
        } else {
          return SYNTHSR_EINVAL;
        }
      }
      if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
      if (KS && (bias != nullptr || act != 0)) {
        hipLaunchKernelGGL(bias_act_kernel, dim3(syn_grid(nout, 256)), dim3(256), 0, st, out, bias, nout, Cout, act,
                         act == 2 ? ext.addend : nullptr);
        if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
      }
      return SYNTHSR_OK;
    }
  }
  static SynOncePerDevice attr_done;
  auto kern = conv3d_fwd_kernel<CK, NT, MT, KS, NV>;
  if (auto once_ = attr_done.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  const int64_t nout = (int64_t)s[0] * s[1] * s[2] * Cout;
  if (KS) {
    if (act != 2 && ext.addend && ext.addend != out) return SYNTHSR_EINVAL;
    if ((act == 2 || !ext.addend) && hipMemsetAsync(out, 0, (size_t)nout * sizeof(float), st) != hipSuccess)
      return SYNTHSR_ELAUNCH;
  }
  const int gz = KS ? pl.ksplit : (ext.mode == 1 ? 8 : 1);
  hipLaunchKernelGGL(kern, dim3(tiles0 * tiles1 * tiles2, pl.nchunks, gz), dim3(256), smem, st, in, wp, bias, out, s[0],
                     s[1], s[2], Cin, Cout, pl.ncc, tiles1, tiles2, act | (PLAN_DBG << 8), ext);
  if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
  if (KS && (bias != nullptr || act != 0)) {
    hipLaunchKernelGGL(bias_act_kernel, dim3(syn_grid(nout, 256)), dim3(256), 0, st, out, bias, nout, Cout, act,
                         act == 2 ? ext.addend : nullptr);
    if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
  }
  return SYNTHSR_OK;
}

template <int NT>
int launch_fwd_persist(const float* in, const float* wp, const float* bias, float* out, const int s[3], int Cin, int Cout,
                       const FwdPlan& pl, int act, hipStream_t st, const float* addend) {
  const int tiles0 = cdiv(s[0], FT0), tiles1 = cdiv(s[1], 4), tiles2 = cdiv(s[2], FT2);
  const int ntiles = tiles0 * tiles1 * tiles2;
  const size_t smem = (size_t)FH0 * 6 * FH2 * 28 * sizeof(float);
  static SynOncePerDevice attr_done;
  auto kern = conv3d_fwd_persist_kernel<NT>;
  if (auto once_ = attr_done.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  int gx = 512 / pl.nchunks;  // 2 workgroups per CU in total
  gx = std::max(8, (gx / 8) * 8);
  const int64_t nitems = (int64_t)ntiles * pl.ncc;
  while (gx > 8 && gx > nitems) gx -= 8;
  hipLaunchKernelGGL(kern, dim3(gx, pl.nchunks), dim3(256), smem, st, in, wp, bias, out, s[0], s[1], s[2], Cin, Cout,
                     pl.ncc, tiles1, tiles2, ntiles, act, addend);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

int launch_fwd_p4(const float* in, const float* wp, const float* bias, float* out, const int s[3], int Cin,
                  const FwdPlan& pl, int act, hipStream_t st, const float* addend, float* stats = nullptr) {
  const int tiles0 = cdiv(s[0], FT0), tiles1 = cdiv(s[1], 4), tiles2 = cdiv(s[2], FT2);
  const int ntiles = tiles0 * tiles1 * tiles2;
  const size_t smem = (size_t)FH0 * 6 * FH2 * 28 * sizeof(float);
  static SynOncePerDevice attr_done;
  if (auto once_ = attr_done.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_fwd_p4_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  int gx = 512;
  while (gx > 8 && gx > ntiles) gx -= 8;
  float* partial = nullptr;
  if (stats) {  // BatchNorm statistics of the output: per-workgroup partials in library scratch, then a tiny reduction
    partial = ctx_scratch((size_t)gx * 48 * sizeof(float));
    if (!partial) return SYNTHSR_EWORKSPACE;
  }
  hipLaunchKernelGGL(conv3d_fwd_p4_kernel, dim3(gx), dim3(256), smem, st, in, wp, bias, out, s[0], s[1], s[2], Cin, pl.ncc,
                     tiles1, tiles2, ntiles, act | (PLAN_DBG << 8), addend, partial);
  if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
  if (stats) return synthsr_bn_stats_from_partials(partial, gx, (int64_t)s[0] * s[1] * s[2], 24, stats, st);
  return SYNTHSR_OK;
}

template <int NT, int WM, int WN, bool KS>
int launch_fwd_brick(const float* in, const float* wp, const float* bias, float* out, const int s[3], int Cin, int Cout,
                     const FwdPlan& pl, int act, hipStream_t st, const ConvExt& ext) {
  const int tiles0 = cdiv(s[0], 4), tiles1 = cdiv(s[1], 4 * WM), tiles2 = cdiv(s[2], 4);
  const size_t smem = (size_t)6 * (4 * WM + 2) * 6 * 28 * sizeof(float);
  const int64_t nout = (int64_t)s[0] * s[1] * s[2] * Cout;
  const float* addend = ext.addend;
  if (KS) {
    if (act != 2 && addend && addend != out) return SYNTHSR_EINVAL;
    if ((act == 2 || !addend) && hipMemsetAsync(out, 0, (size_t)nout * sizeof(float), st) != hipSuccess)
      return SYNTHSR_ELAUNCH;
  }
  int gz = KS ? pl.ksplit : (ext.mode == 1 ? 8 : 1);
  if (!KS && ext.mode == 2) {
    gz = parity_split((int64_t)tiles0 * tiles1 * tiles2 * (pl.nchunks / WN), bias, act, ext);
    if (gz > 1 && hipMemsetAsync(out, 0, (size_t)nout * sizeof(float), st) != hipSuccess) return SYNTHSR_ELAUNCH;
  }
  const dim3 grid(tiles0 * tiles1 * tiles2, pl.nchunks / WN, gz);
  if (ext.mode == 0) {
    hipLaunchKernelGGL((conv3d_fwd_brick_kernel<NT, WM, WN, KS, 27>), grid, dim3(64 * WM * WN), smem, st, in, wp, bias, out,
                       s[0], s[1], s[2], Cin, Cout, pl.ncc, tiles1, tiles2, act, ext);
  } else {
    if constexpr (!KS) {
      hipLaunchKernelGGL((conv3d_fwd_brick_kernel<NT, WM, WN, false, 8>), grid, dim3(64 * WM * WN), smem, st, in, wp, bias,
                         out, s[0], s[1], s[2], Cin, Cout, pl.ncc, tiles1, tiles2, act, ext);
    } else {
      return SYNTHSR_EINVAL;
    }
  }
  if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
  if (KS && (bias != nullptr || act != 0)) {
    hipLaunchKernelGGL(bias_act_kernel, dim3(syn_grid(nout, 256)), dim3(256), 0, st, out, bias, nout, Cout, act,
                       act == 2 ? addend : nullptr);
    if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
  }
  return SYNTHSR_OK;
}

template <int NT, int WM, int WN>
int dispatch_fwd_brick2(const float* in, const float* wp, const float* bias, float* out, const int s[3], int Cin, int Cout,
                        const FwdPlan& pl, int act, hipStream_t st, const ConvExt& ext) {
  return pl.ksplit > 1 ? launch_fwd_brick<NT, WM, WN, true>(in, wp, bias, out, s, Cin, Cout, pl, act, st, ext)
                       : launch_fwd_brick<NT, WM, WN, false>(in, wp, bias, out, s, Cin, Cout, pl, act, st, ext);
}

template <int NT>
int dispatch_fwd_brick(const float* in, const float* wp, const float* bias, float* out, const int s[3], int Cin, int Cout,
                       const FwdPlan& pl, int act, hipStream_t st, const ConvExt& ext) {
  if (pl.wn == 4) return dispatch_fwd_brick2<NT, 1, 4>(in, wp, bias, out, s, Cin, Cout, pl, act, st, ext);
  if (pl.wn == 2)
    return pl.wm == 2 ? dispatch_fwd_brick2<NT, 2, 2>(in, wp, bias, out, s, Cin, Cout, pl, act, st, ext)
                      : dispatch_fwd_brick2<NT, 1, 2>(in, wp, bias, out, s, Cin, Cout, pl, act, st, ext);
  return pl.wm == 2 ? dispatch_fwd_brick2<NT, 2, 1>(in, wp, bias, out, s, Cin, Cout, pl, act, st, ext)
                    : dispatch_fwd_brick2<NT, 1, 1>(in, wp, bias, out, s, Cin, Cout, pl, act, st, ext);
}

int launch_fwd_c2(const float* in, const float* wp, const float* bias, float* out, const int s[3], int Cin, int act,
                  hipStream_t st) {
  const int tiles0 = cdiv(s[0], FT0), tiles1 = cdiv(s[1], 4), tiles2 = cdiv(s[2], FT2);
  const int ntiles = tiles0 * tiles1 * tiles2;
  int gx = 2048;  // 8 workgroups per CU: the kernel is bound by its output stores, not by the matrix cores
  while (gx > 8 && gx > ntiles) gx -= 8;
  if (Cin == 2)
    hipLaunchKernelGGL(conv3d_fwd_c2_kernel<2>, dim3(gx), dim3(256), 0, st, in, wp, bias, out, s[0], s[1], s[2], tiles1,
                       tiles2, ntiles, act);
  else
    hipLaunchKernelGGL(conv3d_fwd_c2_kernel<1>, dim3(gx), dim3(256), 0, st, in, wp, bias, out, s[0], s[1], s[2], tiles1,
                       tiles2, ntiles, act);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

int launch_up_fwd_p4(const float* in, const float* wp, const float* bias, float* out, const int s[3], int Cin,
                     const FwdPlan& pl, int act, hipStream_t st, int64_t wstride, const float* addend) {
  const int tiles0 = cdiv(s[0], FT0), tiles1 = cdiv(s[1], 4), tiles2 = cdiv(s[2], FT2);
  const int ntiles = tiles0 * tiles1 * tiles2;
  const size_t smem = (size_t)FH0 * 6 * FH2 * 28 * sizeof(float);
  static SynOncePerDevice attr_done;
  if (auto once_ = attr_done.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_up_fwd_p4_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  int gx = 512;
  while (gx > 8 && gx > ntiles) gx -= 8;
  hipLaunchKernelGGL(conv3d_up_fwd_p4_kernel, dim3(gx), dim3(256), smem, st, in, wp, bias, out, s[0], s[1], s[2], Cin,
                     pl.ncc, tiles1, tiles2, ntiles, act, wstride, addend);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

template <int CK, int NT>
int dispatch_fwd2(const float* in, const float* wp, const float* bias, float* out, const int s[3], int Cin, int Cout,
                  const FwdPlan& pl, int act, hipStream_t st, const ConvExt& ext) {
  if constexpr (CK == 24 && NT <= 3) {
    if (pl.brick) return dispatch_fwd_brick<NT>(in, wp, bias, out, s, Cin, Cout, pl, act, st, ext);
  }
  if (pl.c2) {
    if (ext.mode != 0 || ext.addend) return SYNTHSR_EINVAL;
    return launch_fwd_c2(in, wp, bias, out, s, Cin, act, st);
  }
  if (pl.p4 && ext.mode == 1) {
    if (pl.count() * 32 >= (1ll << 31)) return SYNTHSR_EINVAL;
    return launch_up_fwd_p4(in, wp, bias, out, s, Cin, pl, act, st, ext.wstride, ext.addend);
  }
  if (pl.p4) {
    if ((int64_t)s[0] * s[1] * s[2] * Cin * 4 >= (1ll << 31) || ext.mode != 0) return SYNTHSR_EINVAL;
    return launch_fwd_p4(in, wp, bias, out, s, Cin, pl, act, st, ext.addend);
  }
  if (pl.mt == 4) {
    if constexpr (CK == 24 && NT <= 4) {
      if (pl.nv == 8) return launch_fwd<CK, NT, 4, false, 8>(in, wp, bias, out, s, Cin, Cout, pl, act, st, ext);
    }
    if constexpr (CK == 24 && NT <= 3) {
      if (pl.persist && ext.mode == 0) return launch_fwd_persist<NT>(in, wp, bias, out, s, Cin, Cout, pl, act, st, ext.addend);
    }
    return launch_fwd<CK, NT, 4, false>(in, wp, bias, out, s, Cin, Cout, pl, act, st, ext);
  }
  if constexpr (NT <= 3) {
    if (pl.ksplit > 1) return launch_fwd<CK, NT, 2, true>(in, wp, bias, out, s, Cin, Cout, pl, act, st, ext);
    return launch_fwd<CK, NT, 2, false>(in, wp, bias, out, s, Cin, Cout, pl, act, st, ext);
  }
  return SYNTHSR_EINVAL;
}

template <int CK>
int dispatch_fwd(const float* in, const float* wp, const float* bias, float* out, const int s[3], int Cin, int Cout,
                 const FwdPlan& pl, int act, hipStream_t st, const ConvExt& ext) {
  switch (pl.nt) {
    case 1: return dispatch_fwd2<CK, 1>(in, wp, bias, out, s, Cin, Cout, pl, act, st, ext);
    case 2: return dispatch_fwd2<CK, 2>(in, wp, bias, out, s, Cin, Cout, pl, act, st, ext);
    case 3: return dispatch_fwd2<CK, 3>(in, wp, bias, out, s, Cin, Cout, pl, act, st, ext);
    case 4: return dispatch_fwd2<CK, 4>(in, wp, bias, out, s, Cin, Cout, pl, act, st, ext);
    case 5: return dispatch_fwd2<CK, 5>(in, wp, bias, out, s, Cin, Cout, pl, act, st, ext);
    case 6: return dispatch_fwd2<CK, 6>(in, wp, bias, out, s, Cin, Cout, pl, act, st, ext);
  }
  return SYNTHSR_EINVAL;
}

// ---- deterministic weight gradients: private planes + ordered reduction ------------------------------------------------
// Every weight-gradient kernel accumulates, per workgroup, a partial dW whose addresses are each touched by exactly ONE lane
// of the workgroup; what differs from run to run in the default mode is only the order in which the workgroups' atomic adds
// land on the shared dW.  Deterministic mode (synthsr_set_deterministic) gives every workgroup COLUMN (blockIdx.x = the voxel
// split; the other grid dimensions split dW itself) its own zeroed copy of dW (+ dbias) -- "plane" x -- and a second kernel
// adds the planes up in x order.  No serialisation: the chained-ticket flush this replaces cost ~3 us per hand-over,
// 98 instead of 29 ms per 160^3 step; the planes cost one memset + one read of gx * |dW| floats per launch.
static float* det_planes(size_t floats) {
  SynDeviceState& ds = dev_state();   // the planes belong to the device the call runs on
  const unsigned long long need = (unsigned long long)floats * sizeof(float);
  if (need > ds.det_demand_bytes) ds.det_demand_bytes = need;
  return floats <= ds.det_planes_floats ? ds.det_planes : nullptr;   // too small: SYNTHSR_EWORKSPACE, the caller re-registers
}
__global__ __launch_bounds__(256) void det_reduce_kernel(const float* __restrict__ planes, int64_t stride, int gx,
                                                         float* __restrict__ dw, int64_t dw_elems,
                                                         float* __restrict__ dbias, int cout) {
  const int64_t n = dw_elems + (dbias ? cout : 0);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float t = 0.f;
    for (int x = 0; x < gx; ++x) t += planes[(int64_t)x * stride + i];  // fixed order
    if (i < dw_elems) dw[i] += t;
    else dbias[i - dw_elems] += t;
  }
}
// before the launch: redirects dw / dbias to the planes (no-op unless deterministic mode is on)
static int det_prepare_impl(DetRun* d, float** dw, float** dbias, int64_t dw_elems, int cout, int gx, hipStream_t st) {
  d->stride = 0;
  if (!det_on()) return SYNTHSR_OK;
  d->dw = *dw;
  d->dbias = *dbias;
  d->dw_elems = dw_elems;
  d->cout = cout;
  d->gx = gx;
  d->stride = (dw_elems + cout + 3) / 4 * 4;
  d->planes = det_planes((size_t)gx * (size_t)d->stride);
  if (!d->planes) return SYNTHSR_EWORKSPACE;
  if (hipMemsetAsync(d->planes, 0, (size_t)gx * (size_t)d->stride * sizeof(float), st) != hipSuccess) return SYNTHSR_ELAUNCH;
  *dw = d->planes;
  if (*dbias) *dbias = d->planes + dw_elems;
  return SYNTHSR_OK;
}
static int det_finish_impl(const DetRun* d, hipStream_t st) {
  if (!d->stride) return SYNTHSR_OK;
  const int64_t n = d->dw_elems + (d->dbias ? d->cout : 0);
  hipLaunchKernelGGL(det_reduce_kernel, dim3(syn_grid(n, 256)), dim3(256), 0, st, d->planes, d->stride, d->gx, d->dw,
                     d->dw_elems, d->dbias, d->cout);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

template <int CK, int NT, int MS, int NTAPS>
int launch_wgrad(const float* in, const float* dout, float* dw, const int s[3], int Cin, int Cout, hipStream_t st,
                 const WgExt& ext0) {
  WgExt ext = ext0;
  DetRun det;
  const int64_t dw_elems = (NTAPS == 8) ? 8 * ext.dwstride : (int64_t)27 * ext.cin_total * Cout;
  const int tiles0 = cdiv(s[0], WT0), tiles1 = cdiv(s[1], WT1), tiles2 = cdiv(s[2], WT2);
  const int ntiles = tiles0 * tiles1 * tiles2;
  const int ncc = cdiv(Cin, CK), nco = cdiv(Cout, NT * 16);
  const int ymul = (NTAPS == 8) ? 8 : MS;
  // 512 workgroups in total = the 2 per CU that fit: every extra workgroup only adds a 27*CK*Cout atomic flush
  int gx = (PLAN_FORCE_MT > 8 ? PLAN_FORCE_MT : 512) / (ncc * nco * ymul);
  if (gx < 1) gx = 1;
  if (gx > ntiles) gx = ntiles;
  const size_t smem = ((size_t)(CK + 1) * WVPX + (size_t)NT * 16 * WVPD) * sizeof(float);
  if constexpr (CK == 24) {
    const int dsc = (NTAPS == 8) ? 8 : 1;
    const int64_t xbytes = (int64_t)s[0] * s[1] * s[2] * Cin * 4, dbytes = (int64_t)s[0] * s[1] * s[2] * dsc * Cout * 4;
    if constexpr (NTAPS == 27) {
      // small deep levels: box tiles that divide the volume exactly (4x4x8 for x = 40, 4x4x4 for x = 20)
      const bool div4 = (s[0] % 4) == 0 && (s[1] % 4) == 0 && (s[2] % 4) == 0 && (s[2] % 16) != 0;
      if (PLAN_BRICK && div4 && (Cout % 4) == 0 && xbytes < (1ll << 31) && dbytes < (1ll << 31) && !(PLAN_DBG & 16)) {
        const bool x8 = (s[2] % 8) == 0;
        const int tx = x8 ? 8 : 4;
        const int bt0 = s[0] / 4, bt1 = s[1] / 4, bt2 = s[2] / tx;
        const int bnt = bt0 * bt1 * bt2;
        int bgx = 512 / (ncc * nco * ymul);
        if (bgx < 1) bgx = 1;
        if (bgx > bnt) bgx = bnt;
        const int hv = 6 * 6 * (tx + 2), tv = 16 * tx;
        const int vpx = hv + ((hv / 2) % 2 == 0 ? 2 : 0) + (hv % 2);
        const size_t bsmem = ((size_t)(CK + 1) * vpx + (size_t)NT * 16 * (tv + 2)) * sizeof(float);
        if (const int rc_ = syn_det_prepare(&det, &dw, &ext.dbias, dw_elems, Cout, bgx, st)) return rc_;
        ext.det_stride = det.stride;
        if (x8) {
          hipLaunchKernelGGL((conv3d_wgrad_box_kernel<NT, MS, 4, 4, 8>), dim3(bgx, ncc * ymul, nco), dim3(256), bsmem, st, in,
                             dout, dw, s[0], s[1], s[2], Cin, Cout, bt0, bt1, bt2, ext);
        } else {
          hipLaunchKernelGGL((conv3d_wgrad_box_kernel<NT, MS, 4, 4, 4>), dim3(bgx, ncc * ymul, nco), dim3(256), bsmem, st, in,
                             dout, dw, s[0], s[1], s[2], Cin, Cout, bt0, bt1, bt2, ext);
        }
        if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
        return syn_det_finish(&det, st);
      }
    }
    if ((Cout % 4) == 0 && xbytes < (1ll << 31) && dbytes < (1ll << 31) && !(PLAN_DBG & 16)) {
      static SynOncePerDevice lean_attr_done;
      auto lkern = conv3d_wgrad_lean_kernel<NT, MS, NTAPS>;
      if (auto once_ = lean_attr_done.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lkern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)smem);
      }
      if (const int rc_ = syn_det_prepare(&det, &dw, &ext.dbias, dw_elems, Cout, gx, st)) return rc_;
      ext.det_stride = det.stride;
      hipLaunchKernelGGL(lkern, dim3(gx, ncc * ymul, nco), dim3(256), smem, st, in, dout, dw, s[0], s[1], s[2], Cin, Cout,
                         tiles0, tiles1, tiles2, ext);
      if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
      return syn_det_finish(&det, st);
    }
  }
  static SynOncePerDevice attr_done;
  auto kern = conv3d_wgrad_kernel<CK, NT, MS, NTAPS>;
  if (auto once_ = attr_done.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  if (ext.dbias) {  // the generic kernel has no dbias row
    hipLaunchKernelGGL(colsum_kernel, dim3(1024), dim3(256), 0, st, dout, (int64_t)s[0] * s[1] * s[2], Cout, ext.dbias);
    if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
  }
  float* no_dbias = nullptr;
  if (const int rc_ = syn_det_prepare(&det, &dw, &no_dbias, dw_elems, Cout, gx, st)) return rc_;
  ext.det_stride = det.stride;
  hipLaunchKernelGGL(kern, dim3(gx, ncc * ymul, nco), dim3(256), smem, st, in, dout, dw, s[0], s[1], s[2], Cin, Cout,
                     tiles0, tiles1, tiles2, ext);
  if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
  return syn_det_finish(&det, st);
}

// ---- weight gradient of a 24-input-channel chunk with Cout = 24 on the 4x4x1 MFMA ----------------------------------
// The 16x16x4 kernel pads Cout = 24 to 32 columns and 648 (tap, ci) rows to 704.  Here (cf. conv3d_wgrad_c2_kernel):
//   A (one block broadcast by ABID) = dz: 8 consecutive voxels x 24 channels = 192 contiguous floats = 3 coalesced
//     registers straight from memory (register r, lane l <-> float 64 r + l);
//   B = x: a register holds 16 blocks of 4 consecutive input channels, block = (tap, channel quad); the lane reads its
//     element of the voxel's neighbourhood from the [voxel][24+4] LDS halo tile (per-lane row offset + immediate);
//   D[i] of lane (block, j) = dW[tap][4 quad + j][4 g + i].
// The 162 blocks are split over the 4 waves (41/41/40/40 -> 3 registers each, 85 % of the MFMA work useful instead of
// 69 %); every wave walks all 256 voxels of the 4x4x16 tile: per voxel 3 ds_read_b32 + 18 MFMAs.  The dz registers of
// the next z-plane and the x halo of the next tile are requested one plane / one tile ahead.
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_p4_kernel(const float* __restrict__ in,
                                                                 const float* __restrict__ dout, float* __restrict__ dw,
                                                                 int D0, int D1, int D2, int Cin, int tiles1, int tiles2,
                                                                 int ntiles, int cin_total, int ci_off, int dbg,
                                                                 float* __restrict__ dbias, int64_t det_stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CK = 24, MT = 4, Cout = 24;
  constexpr int FT1 = MT, FH1 = MT + 2, CKP = CK + 4, C4 = CK / 4;
  constexpr int NBLK = 27 * 6, BPW = (NBLK + 3) / 4, NQ = (BPW + 15) / 16;  // 41 blocks per wave in 3 registers
  constexpr uint32_t OOB = 0x80000000u;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cc = blockIdx.y;
  const int G = gridDim.x;
  const int my_pos = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);

  // B rows of this lane: block blk = wave*BPW + 16 q + (lane >> 2), channel 4*quad + (lane & 3)
  int rowoff[NQ];
  bool rowok[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int lb = 16 * q + (lane >> 2);
    const int blk = wave * BPW + lb;
    rowok[q] = lb < BPW && blk <= NBLK;  // block NBLK (a free slot of wave 3) is the dbias row, see below
    const int bb = (lb < BPW && blk < NBLK) ? blk : 0;
    const int tap = bb / 6, quad = bb - tap * 6;
    rowoff[q] = (((tap / 9) * FH1 + (tap / 3) % 3) * FH2 + tap % 3) * CKP + quad * 4 + (lane & 3);
    if (lb < BPW && blk == NBLK) rowoff[q] = ((FH1 + 1) * FH2 + 1) * CKP + CK + (lane & 3);  // centre tap, pad channels
  }
  // pad channels 24..27 of every halo voxel = (1, 0, 0, 0): the staging never writes them, and lane j = 0 of block NBLK
  // reads the 1 as its "x" -> D = sum over voxels of dz = dbias
  static_assert(NBLK - 3 * BPW < BPW && 3 * BPW + BPW > NBLK, "block NBLK must fall into wave 3's slots");
  for (int v = tid; v < FH0 * FH1 * FH2; v += 256)
    *reinterpret_cast<float4*>(&lds[v * CKP + CK]) = make_float4(1.f, 0.f, 0.f, 0.f);

  // ---- x halo staging (as conv3d_fwd_p4_kernel)
  constexpr int PLANE4 = FH1 * FH2 * C4, NJ = (PLANE4 + 255) / 256, NLD = NJ * FH0;
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)((int64_t)D0 * D1 * D2 * Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dout), 0, (int)((int64_t)D0 * D1 * D2 * Cout * 4), 0x00020000);
  int rel[NJ], ldsa[NJ];
  uint32_t cmask[NJ];
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    const int j = tid + 256 * i;
    const int hy = j / (FH2 * C4), r = j - hy * (FH2 * C4), hx = r / C4, c4 = r - hx * C4;
    rel[i] = ((hy * D2 + hx) * Cin + c4 * 4) * 4;
    ldsa[i] = ((hy * FH2 + hx) * CKP + c4 * 4);
    cmask[i] = j < PLANE4 ? ((1u << hy) | (1u << (8 + hx))) : 0xFFFFFFFFu;
  }
  const int plane_bytes = D1 * D2 * Cin * 4;
  int kk[3];  // voxel (0..7) of the octet that dz register r of this lane belongs to
#pragma unroll
  for (int r = 0; r < 3; ++r) kk[r] = (64 * r + lane) / 24;
  float4 stg[NLD];
  auto tile_origin = [&](int t, int& z0, int& y0, int& x0) {
    const int t2 = t % tiles2, t1 = (t / tiles2) % tiles1, t0 = t / (tiles2 * tiles1);
    z0 = t0 * FT0;
    y0 = t1 * FT1;
    x0 = t2 * FT2;
  };
  auto load_halo = [&](int t) {
    int z0, y0, x0;
    tile_origin(t, z0, y0, x0);
    uint32_t bad = 0x80000000u;
#pragma unroll
    for (int h = 0; h < FH1; ++h) bad |= ((unsigned)(y0 - 1 + h) >= (unsigned)D1) ? (1u << h) : 0u;
#pragma unroll
    for (int h = 0; h < FH2; ++h) bad |= ((unsigned)(x0 - 1 + h) >= (unsigned)D2) ? (1u << (8 + h)) : 0u;
    const int yx = (((y0 - 1) * D2 + (x0 - 1)) * Cin + cc * CK) * 4;
    uint32_t voff[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) voff[i] = (cmask[i] & bad) ? OOB : (uint32_t)(rel[i] + yx);
#pragma unroll
    for (int hz = 0; hz < FH0; ++hz) {
      const int gz = z0 - 1 + hz;
      const bool pv = (unsigned)gz < (unsigned)D0;
#pragma unroll
      for (int i = 0; i < NJ; ++i) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)(pv ? voff[i] : OOB), pv ? gz * plane_bytes : 0, 0);
        stg[hz * NJ + i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
    }
  };
  // dz of z-plane `z` of tile (z0, y0, x0): 8 octets (row y = o >> 1, x half o & 1) x 3 registers
  auto load_dz = [&](int z0, int y0, int x0, int z, float (&a)[8][3]) {
    const int gz = z0 + z;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      const int gy = y0 + (o >> 1), xs = x0 + 8 * (o & 1);
      const bool ok = gz < D0 && gy < D1;  // scalar
      const int so = ok ? (((gz * D1 + gy) * D2 + xs) * Cout) * 4 : 0;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int vo = (ok && xs + kk[r] < D2) ? (64 * r + lane) * 4 : (int)OOB;
        a[o][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rdo, vo, so, 0));
      }
    }
  };

  f32x4 acc[NQ][6];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int g = 0; g < 6; ++g) acc[q][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float an[8][3];
  int z0 = 0, y0 = 0, x0 = 0;
  if (my_pos < ntiles) {
    load_halo(my_pos);
    tile_origin(my_pos, z0, y0, x0);
    load_dz(z0, y0, x0, 0, an);
  }
  for (int t = my_pos; t < ntiles; t += G) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      if (i < NJ - 1 || tid + 256 * i < PLANE4) {
#pragma unroll
        for (int hz = 0; hz < FH0; ++hz)
          *reinterpret_cast<float4*>(&lds[ldsa[i] + hz * (FH1 * FH2 * CKP)]) = stg[hz * NJ + i];
      }
    }
    __syncthreads();
    const bool has_next = t + G < ntiles;
    if (has_next) load_halo(t + G);
    int nz0 = z0, ny0 = y0, nx0 = x0;
    if (has_next) tile_origin(t + G, nz0, ny0, nx0);
    for (int z = 0; z < FT0; ++z) {
      float ar[8][3];
#pragma unroll
      for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int r = 0; r < 3; ++r) ar[o][r] = an[o][r];
      if (z + 1 < FT0) {
        load_dz(z0, y0, x0, z + 1, an);
      } else if (has_next) {
        load_dz(nz0, ny0, nx0, 0, an);
      }
      int rb[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) rb[q] = rowoff[q] + z * (FH1 * FH2 * CKP);
      // 64 voxels of the plane; the B elements of voxel s+1 are read while the 18 MFMAs of voxel s issue
      float xq[2][NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) xq[0][q] = lds[rb[q]];
      sfor<0, 64>([&](auto S) {
        constexpr int sv = decltype(S)::value, o = sv / 8, k = sv % 8;
        constexpr int sn = sv + 1 < 64 ? sv + 1 : sv, on = sn / 8, kn = sn % 8;
        constexpr int nbase = ((on >> 1) * FH2 + 8 * (on & 1) + kn) * CKP;
#pragma unroll
        for (int q = 0; q < NQ; ++q) xq[(sv + 1) & 1][q] = lds[rb[q] + nbase];
        __builtin_amdgcn_sched_barrier(0);
        sfor<0, 6>([&](auto GI) {
          constexpr int g = decltype(GI)::value, GG = k * 6 + g;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            acc[q][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(ar[o][GG / 16], xq[sv & 1][q], acc[q][g], 4, GG % 16, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    z0 = nz0;
    y0 = ny0;
    x0 = nx0;
  }
  if (dbg & 8) return;
  // ---- flush through LDS: the 648 x 24 partial of this workgroup is laid out like the dW chunk ([tap][ci][co]), so that
  // every atomic instruction covers 64 consecutive floats (per-lane rows would issue 64 cache-line requests each)
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (!rowok[q]) continue;
    const int blk = wave * BPW + 16 * q + (lane >> 2);  // block NBLK lands right behind the 648 x 24 dW chunk
    float* d = lds + (blk * 4 + (lane & 3)) * Cout;
#pragma unroll
    for (int g = 0; g < 6; ++g) *reinterpret_cast<float4*>(d + 4 * g) = make_float4(acc[q][g][0], acc[q][g][1], acc[q][g][2], acc[q][g][3]);
  }
  __syncthreads();
  const size_t detoff = (size_t)blockIdx.x * det_stride;  // deterministic mode: this workgroup column's own plane (0 otherwise)
  for (int e = tid; e < 27 * CK * Cout; e += 256) {
    const int tap = e / (CK * Cout), r = e - tap * (CK * Cout);
    atomicAdd(dw + detoff + ((size_t)tap * cin_total + ci_off + cc * CK) * Cout + r, lds[e]);
  }
  if (dbias && cc == 0 && tid < Cout) atomicAdd(dbias + detoff + tid, lds[NBLK * 4 * Cout + tid]);
}

int launch_wgrad_c2(const float* in, const float* dout, float* dw, float* dbias, const int s[3], int Cin, hipStream_t st,
                    const WgExt& ext) {
  const int tiles0 = cdiv(s[0], FT0), tiles1 = cdiv(s[1], 4), tiles2 = cdiv(s[2], FT2);
  const int ntiles = tiles0 * tiles1 * tiles2;
  int gx = PLAN_FORCE_MT > 8 ? PLAN_FORCE_MT : 2048;  // 8 per CU: per-tile work is short, latency is hidden by occupancy
  while (gx > 8 && gx > ntiles) gx -= 8;
  float* partial = ctx_scratch((size_t)gx * 1536 * sizeof(float));
  if (!partial) return SYNTHSR_EWORKSPACE;
  if (Cin == 2)
    hipLaunchKernelGGL(conv3d_wgrad_c2_kernel<2>, dim3(gx), dim3(256), 0, st, in, dout, partial, s[0], s[1], s[2], tiles1,
                       tiles2, ntiles);
  else
    hipLaunchKernelGGL(conv3d_wgrad_c2_kernel<1>, dim3(gx), dim3(256), 0, st, in, dout, partial, s[0], s[1], s[2], tiles1,
                       tiles2, ntiles);
  if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
  const int nrow = 27 * Cin;
  hipLaunchKernelGGL(rows_reduce_kernel, dim3(cdiv((nrow + 1) * 24, 64), 32), dim3(64), 0, st, partial, gx, dw, dbias, nrow,
                     Cin, ext.cin_total, ext.ci_off);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

template <int NTAPS>
int dispatch_wgrad(const float* in, const float* dout, float* dw, const int shape[3], int Cin, int Cout, hipStream_t st,
                   const WgExt& ext) {
  if constexpr (NTAPS == 8) {
    // fp32 through three bf16 pieces per operand: the eight waves of a workgroup = the eight parities over one staged x halo
    if (up_wgrad_takes_split(shape, Cin, Cout)) {
      const int rc = syn_split_upwgrad(in, dout, dw, shape, Cin, Cout, cfg().nprod, st);
      if (rc != SYNTHSR_EINVAL) return rc;
    }
    const int64_t vox = (int64_t)shape[0] * shape[1] * shape[2];
    const int tiles0 = cdiv(shape[0], FT0), tiles1 = cdiv(shape[1], 4), tiles2 = cdiv(shape[2], FT2);
    const int ntiles = tiles0 * tiles1 * tiles2, ncc = Cin / 24;
    if (Cout == 24 && (Cin % 24) == 0 && PLAN_P4 && ntiles >= 768 && vox * Cin * 4 < (1ll << 31) &&
        vox * 8 * Cout * 4 < (1ll << 31) && !(PLAN_DBG & 16)) {
      const size_t smem = (size_t)FH0 * 6 * FH2 * 28 * sizeof(float);
      static SynOncePerDevice attr_done;
      if (auto once_ = attr_done.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_up_wgrad_p4_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      }
      int gx = std::max(8, ((512 / (ncc * 2)) / 8) * 8);  // each workgroup carries 4 of the 8 parities (one per wave)
      while (gx > 8 && gx > ntiles) gx -= 8;
      DetRun det;
      float* no_dbias = nullptr;
      if (const int rc_ = syn_det_prepare(&det, &dw, &no_dbias, 8 * ext.dwstride, Cout, gx, st)) return rc_;
      hipLaunchKernelGGL(conv3d_up_wgrad_p4_kernel, dim3(gx, ncc * 2), dim3(256), smem, st, in, dout, dw, shape[0], shape[1],
                         shape[2], Cin, tiles1, tiles2, ntiles, ext.dwstride, ext.dbg, det.stride);
      if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
      return syn_det_finish(&det, st);
    }
  }
  if constexpr (NTAPS == 27) {
    if (Cin <= 2 && Cout == 24 && PLAN_P4 && (int64_t)shape[0] * shape[1] * shape[2] * Cout * 4 < (1ll << 31))
      return launch_wgrad_c2(in, dout, dw, ext.dbias, shape, Cin, st, ext);
    // fp32 through three bf16 pieces per operand (conv_split.hip): layers with enough 4x4x16 tiles
    if (wgrad_takes_split(shape, Cin, Cout)) {
      const int rc = syn_split_wgrad(in, dout, dw, ext.dbias, shape, ext.cin_total, ext.ci_off, Cin, Cout, cfg().nprod, st);
      if (rc != SYNTHSR_EINVAL) return rc;  // EINVAL: channel counts the split kernel does not cover
    }
  }
  if constexpr (NTAPS == 27) {
    const int64_t vox = (int64_t)shape[0] * shape[1] * shape[2];
    if (Cout == 24 && (Cin % 24) == 0 && PLAN_P4 && vox * Cin * 4 < (1ll << 31) && vox * Cout * 4 < (1ll << 31) &&
        !(PLAN_DBG & 16)) {
      const int tiles0 = cdiv(shape[0], FT0), tiles1 = cdiv(shape[1], 4), tiles2 = cdiv(shape[2], FT2);
      const int ntiles = tiles0 * tiles1 * tiles2, ncc = Cin / 24;
      const size_t smem = (size_t)FH0 * 6 * FH2 * 28 * sizeof(float);
      static SynOncePerDevice attr_done;
      if (auto once_ = attr_done.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_wgrad_p4_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      }
      int gx = std::max(8, ((512 / ncc) / 8) * 8);
      while (gx > 8 && gx > ntiles) gx -= 8;
      DetRun det;
      float* dbias = ext.dbias;
      if (const int rc_ = syn_det_prepare(&det, &dw, &dbias, (int64_t)27 * ext.cin_total * Cout, Cout, gx, st)) return rc_;
      hipLaunchKernelGGL(conv3d_wgrad_p4_kernel, dim3(gx, ncc), dim3(256), smem, st, in, dout, dw, shape[0], shape[1],
                         shape[2], Cin, tiles1, tiles2, ntiles, ext.cin_total, ext.ci_off, ext.dbg, dbias, det.stride);
      if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
      return syn_det_finish(&det, st);
    }
  }
  const int CK = ck_for(Cin);
  // output-channel chunks of <= 48 (3 n-tiles) keep the accumulators of all taps x 24 ci in registers
  const int nt_all = cdiv(Cout, 16);
  const int nco = cdiv(nt_all, 3);
  const int NT = cdiv(nt_all, nco);
  if (CK == 24) {
    if (NT == 1) return launch_wgrad<24, 1, 1, NTAPS>(in, dout, dw, shape, Cin, Cout, st, ext);
    if (NT == 2) return launch_wgrad<24, 2, 1, NTAPS>(in, dout, dw, shape, Cin, Cout, st, ext);
    if constexpr (NTAPS == 27) return launch_wgrad<24, 3, 2, 27>(in, dout, dw, shape, Cin, Cout, st, ext);
    return launch_wgrad<24, 3, 1, NTAPS>(in, dout, dw, shape, Cin, Cout, st, ext);
  }
  if (NT == 1) return launch_wgrad<8, 1, 1, NTAPS>(in, dout, dw, shape, Cin, Cout, st, ext);
  if (NT == 2) return launch_wgrad<8, 2, 1, NTAPS>(in, dout, dw, shape, Cin, Cout, st, ext);
  return launch_wgrad<8, 3, 1, NTAPS>(in, dout, dw, shape, Cin, Cout, st, ext);
}

}  // namespace

extern "C" {

int64_t synthsr_conv3d_pack_ex(const synthsr_conv_ctx* ctx, const float* w, float* packed, const int shape[3], int Cin_total,
                               int ci_off, int Cin, int Cout, int mode, int up, synthsr_stream_t stream) {
  const CtxScope scope(ctx);
  if (!scope.ok) return SYNTHSR_EINVAL;
  if (!shape || Cin < 1 || Cout < 1 || ci_off < 0 || ci_off + Cin > Cin_total || (mode != 0 && mode != 1) ||
      shape[0] < 1 || shape[1] < 1 || shape[2] < 1 || up < 0 || up > 2)
    return SYNTHSR_EINVAL;
  const int CinE = mode ? Cout : Cin, CoutE = mode ? Cin : Cout;
  // up = 1: parity sets of the folded decoder conv (forward -> up_fwd plan, data gradient -> up_dgrad plan);
  // up = 2: parity sets of a stride-2 conv, whose FORWARD runs through up_dgrad and whose data gradient through up_fwd
  const int kind = !up ? 1 : ((up == 1) == (mode == 0) ? 2 : 0);
  const FwdPlan pl = plan_fwd(shape, CinE, CoutE, kind);
  const int64_t per = pl.count();
  if (per >= (1ll << 31)) return SYNTHSR_EINVAL;  // pack_value indexes one weight set with 32-bit arithmetic
  const int64_t total = per * (up ? 8 : 1);
  if (!packed) return total;
  if (!w) return SYNTHSR_EINVAL;
  for (int p = 0; p < (up ? 8 : 1); ++p) {
    hipLaunchKernelGGL(pack_kernel, dim3(syn_grid(per, 256)), dim3(256), 0, (hipStream_t)stream, w, packed + p * per,
                       Cin_total, ci_off, Cin, Cout, mode, pl.ck, pl.ncc, pl.pack_nt(), pl.nchunks,
                       up ? p + (up == 2 ? 8 : 0) : -1, pl.nv, pl.mfma_count(), per);
    if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
  }
  return total;
}

int64_t synthsr_conv3d_pack(const synthsr_conv_ctx* ctx, const float* w, float* packed, const int shape[3], int Cin, int Cout,
                            int mode, synthsr_stream_t stream) {
  return synthsr_conv3d_pack_ex(ctx, w, packed, shape, Cin, 0, Cin, Cout, mode, 0, stream);
}

int synthsr_conv3d_plan(const synthsr_conv_ctx* ctx, const int shape[3], int CinE, int CoutE, int plain, int64_t out[8]) {
  const CtxScope scope(ctx);
  if (!scope.ok) return SYNTHSR_EINVAL;
  if (!shape || !out || CinE < 1 || CoutE < 1 || shape[0] < 1 || shape[1] < 1 || shape[2] < 1) return SYNTHSR_EINVAL;
  const FwdPlan pl = plan_fwd(shape, CinE, CoutE, plain);
  out[0] = pl.ck;
  out[1] = pl.ncc;
  out[2] = pl.pack_nt();
  out[3] = pl.nchunks;
  out[4] = pl.mt;
  out[5] = (pl.split == 1 && syn_split_fwd_halves(shape, CinE, pl.nchunks, pl.stacked, cfg().nprod)) ? 2 : pl.ksplit;
  out[6] = pl.nv;
  out[7] = pl.count();
  return SYNTHSR_OK;
}

int synthsr_conv3d_wgrad_runs_split(const synthsr_conv_ctx* ctx, const int shape[3], int Cin, int Cout) {
  const CtxScope scope(ctx);
  if (!scope.ok) return SYNTHSR_EINVAL;
  if (!shape || Cin < 1 || Cout < 1 || shape[0] < 1 || shape[1] < 1 || shape[2] < 1) return SYNTHSR_EINVAL;
  return (Cin > 2 || Cout != 24) && wgrad_takes_split(shape, Cin, Cout) ? 1 : 0;
}

int synthsr_conv3d_up_wgrad_runs_split(const synthsr_conv_ctx* ctx, const int lo_shape[3], int Cl, int Cout) {
  const CtxScope scope(ctx);
  if (!scope.ok) return SYNTHSR_EINVAL;
  if (!lo_shape || Cl < 1 || Cout < 1 || lo_shape[0] < 1 || lo_shape[1] < 1 || lo_shape[2] < 1) return SYNTHSR_EINVAL;
  return up_wgrad_takes_split(lo_shape, Cl, Cout) ? 1 : 0;
}

int synthsr_conv3d_pack_all(const float* params, float* packed, const int64_t* jobs_dev, int njobs,
                            synthsr_stream_t stream) {
  if (!params || !packed || !jobs_dev || njobs < 1) return SYNTHSR_EINVAL;
  // 256 workgroups per job: the few large layers (384 -> 384: 4 M elements per copy) set the duration, 64 workgroups each
  // left three quarters of the CUs idle for most of the kernel's 220 us
  hipLaunchKernelGGL(pack_all_kernel, dim3(256, njobs), dim3(256), 0, (hipStream_t)stream, params, packed, jobs_dev);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

// weight gradient of a stride-2 conv from the per-parity partials of synthsr_conv3d_up_wgrad called with lo := dy (Co
// channels) and dout := x (the 2x tensor, Ci channels): dwc [8][27][Co][Ci], dw [27][Ci][Co] += .  Tap t of an axis sits at
// parity p = t & 1, slot 1 - (t >> 1)  (t 0 -> (0, 1), t 1 -> (1, 1), t 2 -> (0, 0)).
__global__ void stride_unpack_kernel(const float* __restrict__ dwc, float* __restrict__ dw, int Ci, int Co, int64_t total) {
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(idx % Co);
    const int ci = (int)((idx / Co) % Ci);
    const int t = (int)(idx / ((int64_t)Co * Ci));
    const int tz = t / 9, ty = (t / 3) % 3, tx = t % 3;
    const int p = ((tz & 1) << 2) | ((ty & 1) << 1) | (tx & 1);
    const int slot = ((1 - (tz >> 1)) * 3 + (1 - (ty >> 1))) * 3 + (1 - (tx >> 1));
    dw[idx] += dwc[(((int64_t)p * 27 + slot) * Co + co) * Ci + ci];
  }
}

int synthsr_conv3d_stride_unpack(const float* dwc, float* dw, int Ci, int Co, synthsr_stream_t stream) {
  if (!dwc || !dw || Ci < 1 || Co < 1) return SYNTHSR_EINVAL;
  const int64_t total = (int64_t)27 * Ci * Co;
  hipLaunchKernelGGL(stride_unpack_kernel, dim3(syn_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, dwc, dw, Ci, Co,
                     total);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

int synthsr_conv3d_up_unpack(float* dwc, float* dw, int Cin_total, int ci_off, int Cl, int Cout, synthsr_stream_t stream) {
  if (!dwc || !dw || Cl < 1 || Cout < 1 || ci_off < 0 || ci_off + Cl > Cin_total) return SYNTHSR_EINVAL;
  const int64_t total = (int64_t)27 * Cl * Cout;
  hipLaunchKernelGGL(up_unpack_kernel, dim3(syn_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, dwc, dw, Cin_total,
                     ci_off, Cl, Cout, total);
  if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
  const int64_t total64 = (int64_t)64 * Cl * Cout;  // the partials are consumed: dwc is all zeros again
  hipLaunchKernelGGL(up_clear_kernel, dim3(syn_grid(total64, 256)), dim3(256), 0, (hipStream_t)stream, dwc, Cl, Cout, total64);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

int synthsr_conv3d_fwd(const synthsr_conv_ctx* ctx, const float* in, const float* wpacked, const float* bias, float* out,
                       const int shape[3], int Cin, int Cout, int act, synthsr_stream_t stream) {
  const CtxScope scope(ctx);
  if (!scope.ok) return SYNTHSR_EINVAL;
  if (!in || !wpacked || !out || !shape || Cin < 1 || Cout < 1 || shape[0] < 1 || shape[1] < 1 || shape[2] < 1 ||
      (act != 0 && act != 1))
    return SYNTHSR_EINVAL;
  const FwdPlan pl = plan_fwd(shape, Cin, Cout);
  if (pl.split)
    return syn_split_fwd(in, wpacked, bias, nullptr, out, shape, Cin, Cout, pl.mt, pl.nchunks, act, nullptr, nullptr, 0,
                         pl.stacked, cfg().nprod, (hipStream_t)stream);
  const ConvExt ext{0, nullptr, 0, pl.mfma_count()};
  if (pl.ck == 24) return dispatch_fwd<24>(in, wpacked, bias, out, shape, Cin, Cout, pl, act, (hipStream_t)stream, ext);
  if (pl.ck == 32) return dispatch_fwd<32>(in, wpacked, bias, out, shape, Cin, Cout, pl, act, (hipStream_t)stream, ext);
  return dispatch_fwd<8>(in, wpacked, bias, out, shape, Cin, Cout, pl, act, (hipStream_t)stream, ext);
}

int synthsr_conv3d_fwd_add(const synthsr_conv_ctx* ctx, const float* in, const float* wpacked, const float* bias,
                           const float* addend, float* out, const int shape[3], int Cin, int Cout, int act,
                           synthsr_stream_t stream) {
  const CtxScope scope(ctx);
  if (!scope.ok) return SYNTHSR_EINVAL;
  if (!in || !wpacked || !out || !shape || Cin < 1 || Cout < 1 || shape[0] < 1 || shape[1] < 1 || shape[2] < 1 ||
      (act != 0 && act != 1 && act != 2) || (act == 2 && (!addend || addend == out)))
    return SYNTHSR_EINVAL;
  const FwdPlan pl = plan_fwd(shape, Cin, Cout);
  if (pl.split)
    return syn_split_fwd(in, wpacked, bias, addend, out, shape, Cin, Cout, pl.mt, pl.nchunks, act, nullptr, nullptr, 0,
                         pl.stacked, cfg().nprod, (hipStream_t)stream);
  const ConvExt ext{0, addend, 0, pl.mfma_count()};
  if (pl.ck == 24) return dispatch_fwd<24>(in, wpacked, bias, out, shape, Cin, Cout, pl, act, (hipStream_t)stream, ext);
  if (pl.ck == 32) return dispatch_fwd<32>(in, wpacked, bias, out, shape, Cin, Cout, pl, act, (hipStream_t)stream, ext);
  return dispatch_fwd<8>(in, wpacked, bias, out, shape, Cin, Cout, pl, act, (hipStream_t)stream, ext);
}

int synthsr_conv3d_fwd_stats(const synthsr_conv_ctx* ctx, const float* in, const float* wpacked, const float* bias, float* out,
                             const int shape[3], int Cin, int Cout, int act, float* stats, double* ws, synthsr_stream_t stream) {
  const CtxScope scope(ctx);
  if (!scope.ok) return SYNTHSR_EINVAL;
  if (!in || !wpacked || !out || !shape || !stats || !ws || Cin < 1 || Cout < 1 || shape[0] < 1 || shape[1] < 1 ||
      shape[2] < 1 || (act != 0 && act != 1))
    return SYNTHSR_EINVAL;
  const int64_t nvox = (int64_t)shape[0] * shape[1] * shape[2];
  const FwdPlan pl = plan_fwd(shape, Cin, Cout);
  if (pl.split) {  // statistics accumulated in the conv epilogue (conv_split.hip)
    float* partial = ctx_scratch((size_t)512 * 2 * Cout * sizeof(float));
    if (!partial) return SYNTHSR_EWORKSPACE;
    return syn_split_fwd(in, wpacked, bias, nullptr, out, shape, Cin, Cout, pl.mt, pl.nchunks, act, stats, partial, 0,
                         pl.stacked, cfg().nprod, (hipStream_t)stream);
  }
  if (pl.p4 && nvox * Cin * 4 < (1ll << 31))  // statistics accumulated in the conv epilogue
    return launch_fwd_p4(in, wpacked, bias, out, shape, Cin, pl, act, (hipStream_t)stream, nullptr, stats);
  const int rc = synthsr_conv3d_fwd(ctx, in, wpacked, bias, out, shape, Cin, Cout, act, stream);
  if (rc != SYNTHSR_OK) return rc;
  return synthsr_bn_stats(out, nvox, Cout, stats, ws, stream);
}

int synthsr_conv3d_up_fwd(const synthsr_conv_ctx* ctx, const float* lo, const float* wpacked8, const float* bias,
                          const float* addend, float* out, const int lo_shape[3], int Cl, int Cout, int act,
                          synthsr_stream_t stream) {
  const CtxScope scope(ctx);
  if (!scope.ok) return SYNTHSR_EINVAL;
  if (!lo || !wpacked8 || !out || !lo_shape || Cl < 1 || Cout < 1 || lo_shape[0] < 1 || lo_shape[1] < 1 ||
      lo_shape[2] < 1 || (act != 0 && act != 1))
    return SYNTHSR_EINVAL;
  const FwdPlan pl = plan_fwd(lo_shape, Cl, Cout, 2);
  if (pl.split)  // all parities from one converted low-resolution halo (conv_split.hip: conv3d_split_upfwd_kernel)
    return syn_split_upfwd(lo, wpacked8, bias, addend, out, lo_shape, Cl, Cout, pl.mt, act, cfg().nprod, (hipStream_t)stream);
  const int64_t wstride = pl.count();
  const ConvExt ext{1, addend, wstride, pl.mfma_count()};
  if (pl.ck == 24) return dispatch_fwd<24>(lo, wpacked8, bias, out, lo_shape, Cl, Cout, pl, act, (hipStream_t)stream, ext);
  if (pl.ck == 32) return dispatch_fwd<32>(lo, wpacked8, bias, out, lo_shape, Cl, Cout, pl, act, (hipStream_t)stream, ext);
  return dispatch_fwd<8>(lo, wpacked8, bias, out, lo_shape, Cl, Cout, pl, act, (hipStream_t)stream, ext);
}

int synthsr_conv3d_up_dgrad(const synthsr_conv_ctx* ctx, const float* dout, const float* wpacked8, float* dlo,
                            const int lo_shape[3], int Cl, int Cout, synthsr_stream_t stream) {
  const CtxScope scope(ctx);
  if (!scope.ok) return SYNTHSR_EINVAL;
  if (!dout || !wpacked8 || !dlo || !lo_shape || Cl < 1 || Cout < 1 || lo_shape[0] < 1 || lo_shape[1] < 1 ||
      lo_shape[2] < 1)
    return SYNTHSR_EINVAL;
  // effective conv: input channels = Cout (of the forward layer), output channels = Cl
  const FwdPlan pl = plan_fwd(lo_shape, Cout, Cl, 0);
  if (pl.split)  // the 8 parities as K chunks of one split-arithmetic launch (conv_split.hip, UPM 2)
    return syn_split_fwd(dout, wpacked8, nullptr, nullptr, dlo, lo_shape, Cout, Cl, pl.mt, pl.nchunks, 0, nullptr, nullptr, 2, 0,
                         cfg().nprod, (hipStream_t)stream);
  const int64_t wstride = pl.count();
  const ConvExt ext{2, nullptr, wstride, pl.mfma_count()};
  if (pl.ck == 24) return dispatch_fwd<24>(dout, wpacked8, nullptr, dlo, lo_shape, Cout, Cl, pl, 0, (hipStream_t)stream, ext);
  if (pl.ck == 32) return dispatch_fwd<32>(dout, wpacked8, nullptr, dlo, lo_shape, Cout, Cl, pl, 0, (hipStream_t)stream, ext);
  return dispatch_fwd<8>(dout, wpacked8, nullptr, dlo, lo_shape, Cout, Cl, pl, 0, (hipStream_t)stream, ext);
}


// ---- deterministic mode (see common.h: syn_det_gather / syn_turn_begin) ------------------------------------------------
extern "C" __attribute__((visibility("hidden"))) int syn_det_enabled() { return det_on(); }
extern "C" __attribute__((visibility("hidden"))) int syn_det_prepare(DetRun* d, float** dw, float** dbias, int64_t dw_elems,
                                                                      int cout, int gx, hipStream_t st) {
  return det_prepare_impl(d, dw, dbias, dw_elems, cout, gx, st);
}
extern "C" __attribute__((visibility("hidden"))) int syn_det_finish(const DetRun* d, hipStream_t st) {
  return det_finish_impl(d, st);
}
extern "C" int syn_det_set_pointwise(SynDet*);
extern "C" int syn_det_set_critic(SynDet*);
extern "C" int syn_det_set_ssim(SynDet*);
extern "C" int syn_det_set_conv_bf16(SynDet*);
constexpr long long DET_SCRATCH_FLOATS = 16ll << 20;  // 64 MB of partial rows (4096 workgroups x 4096 sums)

int synthsr_set_deterministic(int on) {
  if (hipDeviceSynchronize() != hipSuccess) return SYNTHSR_ELAUNCH;  // no kernel may see the switch mid-flight
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur < 0 || cur >= SYN_MAX_DEVICES) return SYNTHSR_EINVAL;  // never alias another device
  SynDeviceState& ds = g_dev[cur];   // the CURRENT device: its state block, its device symbols, its planes
  if (on && !ds.det_state) {
    // all-or-nothing: a half-built state block (struct allocated, scratch not) must never be installed by a later call
    SynDet* state = nullptr;
    float* scratch = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&state), sizeof(SynDet)) != hipSuccess) return SYNTHSR_ELAUNCH;
    bool ok = hipMalloc(reinterpret_cast<void**>(&scratch), DET_SCRATCH_FLOATS * sizeof(float)) == hipSuccess;
    if (ok) {
      SynDet* h = new SynDet();
      h->scratch_floats = DET_SCRATCH_FLOATS;
      h->scratch = scratch;
      ok = hipMemcpy(state, h, sizeof(SynDet), hipMemcpyHostToDevice) == hipSuccess;
      delete h;
    }
    if (!ok) {
      if (scratch) (void)hipFree(scratch);
      (void)hipFree(state);
      return SYNTHSR_ELAUNCH;
    }
    ds.det_state = state;
  }
  if (ds.det_state) {  // fresh tickets / counters / timeout flag (scratch pointer and size stay)
    if (hipMemset(ds.det_state, 0, offsetof(SynDet, scratch_floats)) != hipSuccess) return SYNTHSR_ELAUNCH;
    if (hipMemset(reinterpret_cast<char*>(ds.det_state) + offsetof(SynDet, chain), 0, sizeof(int) * SYN_DET_CHAINS) != hipSuccess)
      return SYNTHSR_ELAUNCH;
  }
  SynDet* p = on ? ds.det_state : nullptr;
  ds.det = on ? 1 : 0;
  if (syn_det_set_conv3d(p) || syn_det_set_pointwise(p) || syn_det_set_critic(p) || syn_det_set_ssim(p) ||
      syn_det_set_conv_bf16(p))
    return SYNTHSR_ELAUNCH;
  return hipDeviceSynchronize() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

int synthsr_set_deterministic_workspace(void* planes, unsigned long long bytes) {
  if (hipDeviceSynchronize() != hipSuccess) return SYNTHSR_ELAUNCH;  // a queued ordered reduction may still read the old planes
  int d = -1;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= SYN_MAX_DEVICES || (bytes && !planes)) return SYNTHSR_EINVAL;
  SynDeviceState& ds = g_dev[d];
  ds.det_planes = static_cast<float*>(planes);
  ds.det_planes_floats = planes ? (size_t)(bytes / sizeof(float)) : 0;
  return SYNTHSR_OK;
}

unsigned long long synthsr_deterministic_workspace_demand(void) { return dev_state().det_demand_bytes; }

unsigned long long synthsr_conv_workspace_bytes(void) { return (unsigned long long)SYN_WORKSPACE_BOUND; }

int synthsr_deterministic_status(void) {
  const SynDeviceState& ds = dev_state();
  if (!ds.det) return 0;
  int v[2] = {0, 0};
  if (hipMemcpy(v, ds.det_state, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return v[1] ? 2 : 1;  // 1: on and every ordered wait completed; 2: on, but a wait timed out (order not guaranteed)
}

int synthsr_conv3d_wgrad_bias(const synthsr_conv_ctx* ctx, const float* in, const float* dout, float* dw, float* dbias,
                              const int shape[3], int Cin_total, int ci_off, int Cin, int Cout, synthsr_stream_t stream) {
  const CtxScope scope(ctx);
  if (!scope.ok) return SYNTHSR_EINVAL;
  if (!in || !dout || !dw || !shape || Cin < 1 || Cout < 1 || ci_off < 0 || ci_off + Cin > Cin_total || shape[0] < 1 ||
      shape[1] < 1 || shape[2] < 1)
    return SYNTHSR_EINVAL;
  const WgExt ext{0, Cin_total, ci_off, 0, PLAN_DBG, dbias};
  return dispatch_wgrad<27>(in, dout, dw, shape, Cin, Cout, (hipStream_t)stream, ext);
}

int synthsr_conv3d_wgrad_ex(const synthsr_conv_ctx* ctx, const float* in, const float* dout, float* dw, const int shape[3],
                            int Cin_total, int ci_off, int Cin, int Cout, synthsr_stream_t stream) {
  return synthsr_conv3d_wgrad_bias(ctx, in, dout, dw, nullptr, shape, Cin_total, ci_off, Cin, Cout, stream);
}

int synthsr_conv3d_wgrad(const synthsr_conv_ctx* ctx, const float* in, const float* dout, float* dw, const int shape[3], int Cin,
                         int Cout, synthsr_stream_t stream) {
  return synthsr_conv3d_wgrad_ex(ctx, in, dout, dw, shape, Cin, 0, Cin, Cout, stream);
}

int synthsr_conv3d_up_wgrad(const synthsr_conv_ctx* ctx, const float* lo, const float* dout, float* dwc, const int lo_shape[3],
                            int Cl, int Cout, synthsr_stream_t stream) {
  const CtxScope scope(ctx);
  if (!scope.ok) return SYNTHSR_EINVAL;
  if (!lo || !dout || !dwc || !lo_shape || Cl < 1 || Cout < 1 || lo_shape[0] < 1 || lo_shape[1] < 1 || lo_shape[2] < 1)
    return SYNTHSR_EINVAL;
  const WgExt ext{1, Cl, 0, (int64_t)27 * Cl * Cout, PLAN_DBG, nullptr};
  return dispatch_wgrad<8>(lo, dout, dwc, lo_shape, Cl, Cout, (hipStream_t)stream, ext);
}

}  // extern "C"
