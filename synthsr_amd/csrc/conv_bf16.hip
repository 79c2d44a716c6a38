// 3x3x3 'same' Conv3D on NDHWC **bf16** activations / weights with fp32 accumulation (gfx950,
// v_mfma_f32_16x16x32_bf16: A[16 x 32], B[32 x 16], D[16 x 16] fp32 in 4 accumulator registers, 16x the fp32 MFMA
// rate).  BASELINE.json configs[3] / [4] ("bf16 with fp32 norm accum", "mixed bf16"); the reference itself is fp32
// Keras (SynthSR/training.py:330-341), parity is against the fp32 oracle with a stated bf16 tolerance.
//
//   forward / data-gradient:  D[co][voxel] += A[co][(tap, ci)] * B[(tap, ci)][voxel]
//       M = output channels (weights are the A operand, pre-packed in fragment order), N = 16 voxels of an x-row, so a
//       lane ends up with 4 consecutive output channels of ONE voxel -> one 8-byte bf16x4 store (fp32 -> bf16 RNE).
//       K runs over (tap, 8-channel group) pairs of a CK-channel chunk, 4 pairs = 32 K-values per MFMA; the B operand
//       of a lane is the 16 contiguous bytes of that pair in the [voxel][CK + pad] LDS halo tile (one ds_read_b128;
//       row stride 80 B = 5 x 16 B -> the 16 lanes of an x-row hit 16 different bank quads).
//       Workgroup = 4 waves = 4x4x16 output voxels, wave w owns the z = w plane (4 x-rows); persistent workgroups walk
//       the tiles XCD-contiguously, the next halo is in flight (registers) while the current one is multiplied.
//       Epilogue: + bias, ELU | multiply by ELU'(below) (data gradient fused with the ELU backward of the layer below),
//       optional per-workgroup BatchNorm partial sums (fp32, from the un-rounded accumulators).
//   weight gradient:          D[(tap, ci)][co] += A[(tap, ci)][voxel] * B[voxel][co]
//       both operands are K(voxel)-major per lane but channel-major in memory: they are read from the natural
//       [voxel][channel] LDS images with the gfx950 transpose read ds_read_b64_tr_b16 (lane i of a 16-lane group
//       supplies the address of 8-byte chunk C_i, lane l receives C_{4j + l/4}[l % 4], j = 0..3; probed on hardware,
//       tools/ubench/tr_probe.hip).  A rows are (tap, channel-quad) blocks, 4 blocks per 16-row tile, so the 27 x CK/4
//       blocks (+ one constant-1 block = dbias) fill the tiles to 99 %; the 4 waves split the row tiles and walk all 256
//       voxels of the tile, 32 per MFMA; fp32 atomics flush once per workgroup.
#include "common.h"
#include <algorithm>
#include <type_traits>
#include <utility>

SYN_DET_SETTER(conv_bf16)

#ifdef SYN_BF16_TIMING
static __device__ long long* g_tm = nullptr;
extern "C" int synthsr_bf16_timing_buffer(long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_tm), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#define TM(i) do { if (g_tm && (blockIdx.x == 0 || blockIdx.x == 300) && blockIdx.y == 0 && tid == 0 && tix < 40) g_tm[((blockIdx.x ? 1 : 0) * 40 + tix) * 8 + (i)] = clock64(); } while (0)
#else
#define TM(i)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef uint16_t bf16_t;

constexpr int TZ = 4, TY = 4, TX = 16, HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HVOX = HZ * HY * HX;
constexpr uint32_t OOB = 0x80000000u;

// ---- tile schedule shared by the forward and the weight-gradient kernel -------------------------------------------------
// Tiles are enumerated (z-slab of SLAB_Z tile planes, y, z within the slab, x): neighbours in all three directions are
// close in the list.  The list is cut into 8 contiguous parts, one per XCD (consecutive workgroup ids are dealt round-robin
// to the XCDs, each with a private L2), and the workgroups of an XCD walk their part with stride G / 8: the ~64 tiles an
// XCD works on at any time form a compact block, so the halo voxels they share are fetched from HBM once (with the plain
// z-major order every 4x4x16 tile pulled its 2.5x halo through on its own: FETCH ~2.5x the tensor).
constexpr int SLAB_Z = 4;
struct TileWalk {
  int pos, end, stride;
};
__device__ __forceinline__ TileWalk tile_walk(int ntiles) {
  const int G = (int)gridDim.x, b = (int)blockIdx.x;
  TileWalk w;
  if (G % 8 == 0) {
    const int per = (ntiles + 7) / 8, k = b & 7;
    w.pos = k * per + (b >> 3);
    w.end = min(ntiles, (k + 1) * per);
    w.stride = G >> 3;
  } else {
    w.pos = b;
    w.end = ntiles;
    w.stride = G;
  }
  return w;
}
__device__ __forceinline__ void tile_decode(int p, int tiles0, int tiles1, int tiles2, int& z0, int& y0, int& x0) {
  const int t12 = tiles1 * tiles2;
  const int s = p / (SLAB_Z * t12), r = p - s * SLAB_Z * t12;
  const int sz = min(SLAB_Z, tiles0 - s * SLAB_Z);
  const int t1 = r / (sz * tiles2), rr = r - t1 * sz * tiles2;
  const int zz = rr / tiles2, t2 = rr - zz * tiles2;
  z0 = (s * SLAB_Z + zz) * TZ;
  y0 = t1 * TY;
  x0 = t2 * TX;
}

// LDS row strides (bytes), odd multiples of 16 B so that the 16 lanes of an x-row hit 16 different bank quads.
// weight gradient: room for CK channels + the constant-1 pad block; forward: no pad needed (24 ch = 48 B = 3 quads)
// (weight gradient: 32 B x odd -- the 8 consecutive x-voxels one LDS cycle serves then start in 8 different 32-byte bank groups)
__host__ __device__ constexpr int rowb_for(int ck) { return ck == 8 ? 32 : 96; }
__host__ __device__ constexpr int rowb_fwd(int ck) { return ck == 8 ? 16 : (ck == 24 ? 48 : 80); }

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

// ELU(alpha = 1) for a bf16 result: 2^(v log2 e) - 1 through v_exp_f32 has an ABSOLUTE error of ~6e-8 (the subtraction
// cancels near 0), i.e. below half a bf16 ulp of the result for |v| > 3e-5 and negligible below -- no polynomial branch as
// in the fp32 kernels: 4 vector-ALU instructions instead of 14 (the epilogue of a 24 -> 24 tile is 32 values per lane)
__device__ __forceinline__ float elu_f(float v) {
  const float e = __builtin_amdgcn_exp2f(v * 1.44269504088896341f) - 1.f;
  return v > 0.f ? v : e;
}
__device__ __forceinline__ float elu_dy(float y) { return y > 0.f ? 1.f : y + 1.f; }

__device__ __forceinline__ uint32_t f2bf(float f) {  // round to nearest even (finite inputs)
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ float bf2f(uint32_t h) { return __uint_as_float(h << 16); }
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// two fp32 -> packed bf16 pair, round to nearest even: the cast lowers to ONE v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  const bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

// ------------------------------------------------------------------------------------------------ weight packing
// forward layout (A fragments): [co-chunk][cc][step][mt][lane 64][8] bf16; lane = (m = lane & 15, g = lane >> 4):
//   pair p = 4 step + g -> tap = p / C8, c8 = p % C8;  value j: W_eff[tap][cc*CK + c8*8 + j][(chunk*MT + mt)*16 + m]
// mode 0: W_eff = w[tap][ci_off + cie][coe]; mode 1 (data gradient): W_eff[tap][cie][coe] = w[26 - tap][ci_off + coe][cie]
// parity >= 0 (0..7 = 4 pz + 2 py + px): one of the 8 weight sets of the nearest-upsample folding, 8 taps (a, b, c) in {0, 1}^3
// (tap = 4a + 2b + c) on the LOW-resolution grid: forward (mode 0) tap (a, b, c) reads halo offset a + p per axis and carries
// the sum of the original taps behind low-res slot s = a + p (syn_up_axis_taps); data gradient (mode 1, channels transposed)
// tap (a, b, c) reads the parity-p sub-lattice of dz at halo offset (1 - p) + a and carries slot s = 1 + p - a.
__device__ __forceinline__ bf16_t pack_bf16_value(const float* __restrict__ w, uint32_t r, int Cin_total, int ci_off, int Cin,
                                                  int Cout, int mode, int CK, int ncc, int MT, int nsteps, int parity) {
  const int C8 = CK / 8;
  const int CinE = mode ? Cout : Cin, CoutE = mode ? Cin : Cout;
  const int j = r & 7;
  r >>= 3;
  const int lane = r & 63;
  r >>= 6;
  const int mt = r % MT;
  r /= MT;
  const int step = r % nsteps;
  r /= nsteps;
  const int cc = r % ncc;
  const int chunk = r / ncc;
  const int m = lane & 15, g = lane >> 4;
  const int p = 4 * step + g;
  const int tap = p / C8, c8 = p - tap * C8;
  const int cie = cc * CK + c8 * 8 + j, coe = (chunk * MT + mt) * 16 + m;
  float v = 0.f;
  if (tap < (parity < 0 ? 27 : 8) && cie < CinE && coe < CoutE) {
    const int ci = (mode ? coe : cie) + ci_off, co = mode ? cie : coe;
    if (parity < 0) {
      const int slot = mode ? 26 - tap : tap;
      v = w[((int64_t)slot * Cin_total + ci) * Cout + co];
    } else {
      const int ab[3] = {tap >> 2, (tap >> 1) & 1, tap & 1}, pp[3] = {(parity >> 2) & 1, (parity >> 1) & 1, parity & 1};
      int t[3][2], n[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) n[i] = syn_up_axis_taps(pp[i], mode ? 1 + pp[i] - ab[i] : ab[i] + pp[i], t[i]);
      for (int a = 0; a < n[0]; ++a)
        for (int b = 0; b < n[1]; ++b)
          for (int c = 0; c < n[2]; ++c)
            v += w[((int64_t)((t[0][a] * 3 + t[1][b]) * 3 + t[2][c]) * Cin_total + ci) * Cout + co];
    }
  }
  return (bf16_t)f2bf(v);
}

__global__ void pack_bf16_kernel(const float* __restrict__ w, bf16_t* __restrict__ packed, int Cin_total, int ci_off,
                                 int Cin, int Cout, int mode, int CK, int ncc, int MT, int nsteps, int parity, int64_t total) {
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x)
    packed[idx] = pack_bf16_value(w, (uint32_t)idx, Cin_total, ci_off, Cin, Cout, mode, CK, ncc, MT, nsteps, parity);
}

// every packed weight set of a network in ONE launch (38 launches of 5 us per training step otherwise): blockIdx.y = job,
// jobs[j] = {w_off, dst_off, total, Cin_total, ci_off, Cin, Cout, mode, CK, ncc, MT, nsteps, parity} (int64 each, offsets in
// elements; parity -1 = plain 27-tap set)
constexpr int BF16_PACK_JOB_FIELDS = 13;
__global__ void pack_bf16_all_kernel(const float* __restrict__ params, bf16_t* __restrict__ packed,
                                     const int64_t* __restrict__ jobs) {
  const int64_t* jb = jobs + (int64_t)blockIdx.y * BF16_PACK_JOB_FIELDS;
  const float* w = params + jb[0];
  bf16_t* dst = packed + jb[1];
  const int64_t total = jb[2];
  const int Cin_total = (int)jb[3], ci_off = (int)jb[4], Cin = (int)jb[5], Cout = (int)jb[6], mode = (int)jb[7], CK = (int)jb[8],
            ncc = (int)jb[9], MT = (int)jb[10], nsteps = (int)jb[11], parity = (int)jb[12];
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x)
    dst[idx] = pack_bf16_value(w, (uint32_t)idx, Cin_total, ci_off, Cin, Cout, mode, CK, ncc, MT, nsteps, parity);
}

struct Bf16Plan {
  int ck, ncc, mt, nchunks, nsteps;
  int64_t count() const { return (int64_t)nchunks * ncc * nsteps * mt * 64 * 8; }
};

inline Bf16Plan plan_bf16(int CinE, int CoutE, int ntaps = 27) {
  Bf16Plan p;
  p.ck = (CinE % 32 == 0) ? 32 : ((CinE % 24 == 0) ? 24 : 8);
  p.ncc = (CinE + p.ck - 1) / p.ck;
  const int mt_all = (CoutE + 15) / 16;
  p.nchunks = (mt_all + 3) / 4;
  p.mt = (mt_all + p.nchunks - 1) / p.nchunks;
  p.nsteps = (ntaps * (p.ck / 8) + 3) / 4;  // ntaps = 8: one parity set of the nearest-upsample folding
  return p;
}

// ------------------------------------------------------------------------------------------------ forward / dgrad
struct FwdArgs {
  const bf16_t* in;
  const bf16_t* wp;
  const float* bias;
  bf16_t* out;
  const bf16_t* below;   // act == 2: ELU output of the layer below, [vox][CoutE]
  float* stats_partial;  // [gridDim.x * gridDim.y][2][16 * MT] per-workgroup sums / sums of squares, or null
  float* partial;        // split-K (small volumes): fp32 [vox][Cout] accumulated with atomics, epilogue in a second kernel
  int D0, D1, D2, Cin, Cout, ncc, tiles1, tiles2, ntiles, act, ksplit;
  float alpha;           // LeakyReLU slope of act 3 / 4 (the WGAN-GP critic)
  int ncc_real;          // UPM == 2: input-channel chunks of dz per parity (ncc = 8 * ncc_real: the 8 parities are K chunks)
};

// WLDS: the whole fragment-ordered weight set of this workgroup's (co-chunk, single ci-chunk) lives in LDS for the life of
// the persistent workgroup (24 -> 24: 42 KB next to the 31 KB halo tile, 2 workgroups per CU) -- no global weight loads
// in the K loop at all; otherwise fragments stream from L2 two K-steps ahead
// UPM: nearest-upsample folding of a decoder's first conv (unet.py; ext/neuron/models.py:426-444 UpSampling3D -> concatenate
// -> Conv3D): the up-sampled channels never exist at full resolution.
//   UPM 1 (forward): D0..D2 = the LOW-resolution grid; blockIdx.z = output parity (pz, py, px); 8-tap parity weights; the
//         tile's outputs go to the voxels 2 v + p of the 2x tensor `out` (raw fp32-accumulated sums rounded to bf16: the
//         skip-channel conv then adds them in its epilogue, act 5);
//   UPM 2 (data gradient): `in` = dz on the 2x grid; the 8 parities are K chunks (cc = parity * ncc_real + chunk): parity p
//         stages the sub-lattice dz[2 v + p] and multiplies by its transposed 8-tap set; output = d(lo) on the low-res grid.
template <int CK, int MT, bool WLDS, int UPM = 0>
__global__ __launch_bounds__(256, 2) void conv3d_bf16_fwd_kernel(const FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int C8 = CK / 8, ROWB = rowb_fwd(CK), NTAPS = UPM ? 8 : 27, NSTEP = (NTAPS * C8 + 3) / 4;
  static_assert(!(WLDS && UPM), "the folded variants stream their weights");
  constexpr int HBYTES = (HVOX * ROWB + 1023) / 1024 * 1024;  // halo image, then (WLDS) the weight fragments
  constexpr int NPIECE = HVOX * C8, NLD = (NPIECE + 255) / 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4;
  const int chunk = blockIdx.y;
  const TileWalk walk = tile_walk(a.ntiles);
  const int tiles0 = a.ntiles / (a.tiles1 * a.tiles2);
  const int D0 = a.D0, D1 = a.D1, D2 = a.D2, Cin = a.Cin, Cout = a.Cout;

  // per-lane LDS byte offset of the (tap, c8) pair of every K-step (relative to the voxel's halo row)
  int koff[NSTEP];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const int p = 4 * s + g;
    const int tap = p / C8, c8 = p - tap * C8;
    const int tz = UPM ? (tap >> 2) : tap / 9, ty = UPM ? ((tap >> 1) & 1) : (tap / 3) % 3, tx = UPM ? (tap & 1) : tap % 3;
    koff[s] = (tap < NTAPS) ? ((tz * HY + ty) * HX + tx) * ROWB + c8 * 16 : 0;  // beyond the taps: weights are zero
  }
  // folded variants: the 2x2x2 window of parity p starts at halo offset p (forward) / 1 - p (data gradient) per axis
  auto par_shift = [&](int par) {
    const int pz = (par >> 2) & 1, py = (par >> 1) & 1, px = par & 1;
    return UPM == 1 ? ((pz * HY + py) * HX + px) * ROWB : (((1 - pz) * HY + (1 - py)) * HX + (1 - px)) * ROWB;
  };
  const int par1 = UPM == 1 ? (int)blockIdx.z : 0;  // forward: this workgroup's output parity
  const int lbase0 = (wave * HY * HX + m) * ROWB;  // voxel (z = wave, y = 0, x = m) of the tile, tap (0, 0, 0)
  int lbase = lbase0 + (UPM == 1 ? par_shift(par1) : 0);

  // staging pieces of this thread: piece j -> halo voxel j / C8, 16-byte group j % C8
  int prel[NLD], plds[NLD];
  uint32_t pmask[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int j = tid + 256 * i;
    const int v = j / C8, c8 = j - v * C8;
    const int hz = v / (HY * HX), r = v - hz * (HY * HX), hy = r / HX, hx = r - hy * HX;
    // UPM 2: the staged tensor lives on the 2x grid, halo voxel h of the low-res tile = voxel 2 h + p of it
    prel[i] = UPM == 2 ? ((2 * hz * (2 * D1) + 2 * hy) * (2 * D2) + 2 * hx) * Cin * 2 + c8 * 16
                       : ((hz * D1 + hy) * D2 + hx) * Cin * 2 + c8 * 16;
    plds[i] = v * ROWB + c8 * 16;
    pmask[i] = j < NPIECE ? ((1u << hz) | (1u << (6 + hy)) | (1u << (12 + hx))) : 0xFFFFFFFFu;  // hx <= 17 -> bit 29
  }
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(a.in), 0, (int)((int64_t)(UPM == 2 ? 8 : 1) * D0 * D1 * D2 * Cin * 2), 0x00020000);
  u32x4 stg[NLD];
  auto tile_origin = [&](int t, int& z0, int& y0, int& x0) { tile_decode(t, tiles0, a.tiles1, a.tiles2, z0, y0, x0); };
  auto load_halo = [&](int t, int cc_) {
    int z0, y0, x0;
    tile_origin(t, z0, y0, x0);
    const int par = UPM == 2 ? cc_ / a.ncc_real : 0;
    const int cc = UPM == 2 ? cc_ - par * a.ncc_real : cc_;
    uint32_t bad = 0x80000000u;
#pragma unroll
    for (int h = 0; h < HZ; ++h) bad |= ((unsigned)(z0 - 1 + h) >= (unsigned)D0) ? (1u << h) : 0u;
#pragma unroll
    for (int h = 0; h < HY; ++h) bad |= ((unsigned)(y0 - 1 + h) >= (unsigned)D1) ? (1u << (6 + h)) : 0u;
#pragma unroll
    for (int h = 0; h < HX; ++h) bad |= ((unsigned)(x0 - 1 + h) >= (unsigned)D2) ? (1u << (12 + h)) : 0u;
    const int base = UPM == 2
        ? ((((2 * (z0 - 1) + ((par >> 2) & 1)) * (2 * D1) + (2 * (y0 - 1) + ((par >> 1) & 1))) * (2 * D2) +
            (2 * (x0 - 1) + (par & 1))) * Cin + cc * CK) * 2
        : ((((z0 - 1) * D1 + (y0 - 1)) * D2 + (x0 - 1)) * Cin + cc * CK) * 2;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const uint32_t vo = (pmask[i] & bad) ? OOB : (uint32_t)(prel[i] + base);
      stg[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)vo, 0, 0);
    }
  };

  // plain: [co-chunk][cc]; UPM 1: 8 such sets back to back, this workgroup uses the one of its parity; UPM 2: per parity
  // [co-chunk][ncc_real] (indexed in the K loop)
  const bf16x8* __restrict__ wfrag = reinterpret_cast<const bf16x8*>(a.wp) +
      (UPM == 2 ? (int64_t)0 : ((int64_t)par1 * gridDim.y + chunk) * a.ncc * NSTEP * MT * 64) + lane;
  if constexpr (WLDS) {  // ncc == 1 (checked by the launcher): one fragment set, copied once
    const u32x4* src = reinterpret_cast<const u32x4*>(a.wp) + (int64_t)chunk * NSTEP * MT * 64;
    for (int i = tid; i < NSTEP * MT * 64; i += 256) *reinterpret_cast<u32x4*>(lds + HBYTES + i * 16) = src[i];
  }
  float bias_r[MT][4];  // this lane's 4 output channels per co-tile (tile-invariant)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = (chunk * MT + mt) * 16 + 4 * g + i;
      bias_r[mt][i] = (a.bias && co < Cout) ? a.bias[co] : 0.f;
    }
  float s1[MT][4], s2[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) s1[mt][i] = s2[mt][i] = 0.f;

  // epilogue through buffer instructions when the output fits 32-bit byte offsets
  const int64_t out_bytes = (int64_t)(UPM == 1 ? 8 : 1) * D0 * D1 * D2 * Cout * 2;
  const bool fast_epi = !a.partial && out_bytes < (1ll << 31) && (!a.stats_partial || a.act <= 1);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)(fast_epi ? out_bytes : 0), 0x00020000);
  const __amdgpu_buffer_rsrc_t rbelow = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(a.below ? a.below : a.out), 0, (int)(fast_epi ? out_bytes : 0), 0x00020000);
  // split-K: slice blockIdx.z of the input-channel chunks (ksplit == 1: all of them)
  const int kz = UPM == 1 ? 0 : (int)blockIdx.z;  // UPM 1: blockIdx.z is the output parity (no split-K)
  const int cc_lo = (int)(((int64_t)kz * a.ncc) / a.ksplit), cc_hi = (int)(((int64_t)(kz + 1) * a.ncc) / a.ksplit);
  if (walk.pos < walk.end) load_halo(walk.pos, cc_lo);
  int tix = -1;
  (void)tix;
  for (int t = walk.pos; t < walk.end; t += walk.stride) {
    ++tix;
    TM(0);
    f32x4 acc[TY][MT];
#pragma unroll
    for (int y = 0; y < TY; ++y)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[y][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int cc = cc_lo; cc < cc_hi; ++cc) {
      __syncthreads();  // everyone is done reading the previous image
      TM(1);
#pragma unroll
      for (int i = 0; i < NLD; ++i)
        if (i < NLD - 1 || tid + 256 * i < NPIECE) *reinterpret_cast<u32x4*>(lds + plds[i]) = stg[i];
      TM(2);
      __syncthreads();
      TM(3);
      if (cc + 1 < cc_hi) {
        load_halo(t, cc + 1);
      } else if (t + walk.stride < walk.end) {
        load_halo(t + walk.stride, cc_lo);
      }
      TM(4);
      const bf16x8* wf;
      if constexpr (UPM == 2) {
        const int par = cc / a.ncc_real, c = cc - par * a.ncc_real;
        wf = wfrag + (((int64_t)par * gridDim.y + chunk) * a.ncc_real + c) * NSTEP * MT * 64;
        lbase = lbase0 + par_shift(par);
      } else {
        wf = wfrag + (int64_t)cc * NSTEP * MT * 64;
      }
      // software pipeline, pinned with sched_barriers (left alone, the scheduler re-uses ONE register set and waits for
      // every LDS read and every weight load right where it is issued): weights two K-steps ahead (global / L2 latency),
      // the halo-tile reads one step ahead (LDS latency), the 4 x MT MFMAs of the current step in between
      bf16x8 wa[3][MT], xb[2][TY];
      auto wload = [&](auto SS, int mt) -> bf16x8 {
        constexpr int ss = decltype(SS)::value;
        if constexpr (WLDS) return *reinterpret_cast<const bf16x8*>(lds + HBYTES + ((ss * MT) * 64 + lane) * 16 + mt * 1024);
        else return wf[(ss * MT + mt) * 64];
      };
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        wa[0][mt] = wload(std::integral_constant<int, 0>{}, mt);
        if constexpr (NSTEP > 1) wa[1][mt] = wload(std::integral_constant<int, 1>{}, mt);
      }
#pragma unroll
      for (int y = 0; y < TY; ++y) xb[0][y] = *reinterpret_cast<const bf16x8*>(lds + lbase + koff[0] + y * (HX * ROWB));
      sfor<0, NSTEP>([&](auto S) {
        constexpr int s = decltype(S)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (s + 2 < NSTEP) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) wa[(s + 2) % 3][mt] = wload(std::integral_constant<int, s + 2>{}, mt);
        }
        if constexpr (s + 1 < NSTEP) {
#pragma unroll
          for (int y = 0; y < TY; ++y)
            xb[(s + 1) & 1][y] = *reinterpret_cast<const bf16x8*>(lds + lbase + koff[s + 1] + y * (HX * ROWB));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int y = 0; y < TY; ++y)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[y][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[s % 3][mt], xb[s & 1][y], acc[y][mt], 0, 0, 0);
      });
      __builtin_amdgcn_sched_barrier(0);
      TM(5);
    }
    // ---- epilogue: lane (m = x, g): channels (chunk*MT + mt)*16 + 4g + i of voxel (z0 + wave, y0 + y, x0 + m)
    int z0, y0, x0;
    tile_origin(t, z0, y0, x0);
    if (fast_epi) {
      // compile-time activation / statistics variants (no branch per piece), 32-bit offsets through buffer instructions
      // whose out-of-range offset drops the store and returns 0 for the load: no bounds arithmetic on 64-bit addresses
      const int gz = z0 + wave, gx = x0 + m;
      const bool zx_ok = gz < D0 && gx < D2;
      // UPM 1: low-res voxel v of parity p -> voxel 2 v + p of the 2x output
      const uint32_t row0 = UPM == 1
          ? (uint32_t)(((2 * gz + ((par1 >> 2) & 1)) * (2 * D1) + (2 * y0 + ((par1 >> 1) & 1))) * (2 * D2) + (2 * gx + (par1 & 1))) *
                (uint32_t)(Cout * 2)
          : (uint32_t)((gz * D1 + y0) * D2 + gx) * (uint32_t)(Cout * 2);  // bytes
      const uint32_t ystep = (uint32_t)((UPM == 1 ? 4 : 1) * D2 * Cout * 2);
      auto epi = [&](auto ACTC, auto STC) {
        constexpr int ACT = decltype(ACTC)::value;
        constexpr bool ST = decltype(STC)::value;
#pragma unroll
        for (int y = 0; y < TY; ++y) {
          const bool vok = zx_ok && (y0 + y) < D1;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const int co = (chunk * MT + mt) * 16 + 4 * g;
            const uint32_t off = (vok && co < Cout) ? row0 + y * ystep + (uint32_t)(co * 2) : OOB;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = acc[y][mt][i] + bias_r[mt][i];
            if constexpr (ACT == 1) {
#pragma unroll
              for (int i = 0; i < 4; ++i) v[i] = elu_f(v[i]);
            } else if constexpr (ACT == 2) {
              const u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(rbelow, (int)off, 0, 0);
              v[0] *= elu_dy(bf2f(b.x & 0xffffu));
              v[1] *= elu_dy(bf2f(b.x >> 16));
              v[2] *= elu_dy(bf2f(b.y & 0xffffu));
              v[3] *= elu_dy(bf2f(b.y >> 16));
            } else if constexpr (ACT == 3) {
#pragma unroll
              for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : a.alpha * v[i];
            } else if constexpr (ACT == 4) {
              const u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(rbelow, (int)off, 0, 0);
              v[0] *= bf2f(b.x & 0xffffu) > 0.f ? 1.f : a.alpha;
              v[1] *= bf2f(b.x >> 16) > 0.f ? 1.f : a.alpha;
              v[2] *= bf2f(b.y & 0xffffu) > 0.f ? 1.f : a.alpha;
              v[3] *= bf2f(b.y >> 16) > 0.f ? 1.f : a.alpha;
            } else if constexpr (ACT == 5) {  // ELU(conv + bias + addend): the other channel range's partial sums (folding)
              const u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(rbelow, (int)off, 0, 0);
              v[0] = elu_f(v[0] + bf2f(b.x & 0xffffu));
              v[1] = elu_f(v[1] + bf2f(b.x >> 16));
              v[2] = elu_f(v[2] + bf2f(b.y & 0xffffu));
              v[3] = elu_f(v[3] + bf2f(b.y >> 16));
            }
            u32x2 o;
            o.x = pack2(v[0], v[1]);
            o.y = pack2(v[2], v[3]);
            __builtin_amdgcn_raw_buffer_store_b64(o, rout, (int)off, 0, 0);
            if constexpr (ST) {  // statistics of what the next layer will read (the bf16-rounded values)
              const float w = off != OOB ? 1.f : 0.f;
              const float r0 = w * bf2f(o.x & 0xffffu), r1 = w * bf2f(o.x >> 16), r2 = w * bf2f(o.y & 0xffffu),
                          r3 = w * bf2f(o.y >> 16);
              s1[mt][0] += r0; s1[mt][1] += r1; s1[mt][2] += r2; s1[mt][3] += r3;
              s2[mt][0] += r0 * r0; s2[mt][1] += r1 * r1; s2[mt][2] += r2 * r2; s2[mt][3] += r3 * r3;
            }
          }
        }
      };
      using T_ = std::true_type;
      using F_ = std::false_type;
      if (a.stats_partial) {
        if (a.act == 1) epi(std::integral_constant<int, 1>{}, T_{});
        else epi(std::integral_constant<int, 0>{}, T_{});
      } else if (a.act == 0) epi(std::integral_constant<int, 0>{}, F_{});
      else if (a.act == 1) epi(std::integral_constant<int, 1>{}, F_{});
      else if (a.act == 2) epi(std::integral_constant<int, 2>{}, F_{});
      else if (a.act == 3) epi(std::integral_constant<int, 3>{}, F_{});
      else if (a.act == 4) epi(std::integral_constant<int, 4>{}, F_{});
      else epi(std::integral_constant<int, 5>{}, F_{});
      continue;
    }
    // generic path (split-K planes, statistics with other activations, tensors of 2 GiB and more)
    const int gz = z0 + wave, gx = x0 + m;
#pragma unroll
    for (int y = 0; y < TY; ++y) {
      const int gy = y0 + y;
      const bool vok = gz < D0 && gy < D1 && gx < D2;
      const int64_t vox = ((int64_t)gz * D1 + gy) * D2 + gx;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int co = (chunk * MT + mt) * 16 + 4 * g;
        if (!vok || co >= Cout) continue;  // Cout is a multiple of 4
        if (a.partial) {  // this K-slice's own fp32 plane: plain stores, summed by the epilogue kernel
          *reinterpret_cast<f32x4*>(a.partial + ((int64_t)kz * a.D0 * D1 * D2 + vox) * Cout + co) = acc[y][mt];
          continue;
        }
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[y][mt][i] + bias_r[mt][i];
        if (a.act == 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = elu_f(v[i]);
        } else if (a.act == 2) {
          const u32x2 b = *reinterpret_cast<const u32x2*>(a.below + vox * Cout + co);
          v[0] *= elu_dy(bf2f(b.x & 0xffffu));
          v[1] *= elu_dy(bf2f(b.x >> 16));
          v[2] *= elu_dy(bf2f(b.y & 0xffffu));
          v[3] *= elu_dy(bf2f(b.y >> 16));
        } else if (a.act == 3) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : a.alpha * v[i];
        } else if (a.act == 4) {
          const u32x2 b = *reinterpret_cast<const u32x2*>(a.below + vox * Cout + co);
          v[0] *= bf2f(b.x & 0xffffu) > 0.f ? 1.f : a.alpha;
          v[1] *= bf2f(b.x >> 16) > 0.f ? 1.f : a.alpha;
          v[2] *= bf2f(b.y & 0xffffu) > 0.f ? 1.f : a.alpha;
          v[3] *= bf2f(b.y >> 16) > 0.f ? 1.f : a.alpha;
        } else if (a.act == 5) {
          const u32x2 b = *reinterpret_cast<const u32x2*>(a.below + vox * Cout + co);
          v[0] = elu_f(v[0] + bf2f(b.x & 0xffffu));
          v[1] = elu_f(v[1] + bf2f(b.x >> 16));
          v[2] = elu_f(v[2] + bf2f(b.y & 0xffffu));
          v[3] = elu_f(v[3] + bf2f(b.y >> 16));
        }
        u32x2 o;
        o.x = pack2(v[0], v[1]);
        o.y = pack2(v[2], v[3]);
        *reinterpret_cast<u32x2*>(a.out + vox * Cout + co) = o;
        if (a.stats_partial) {  // statistics of what the next layer will read (the bf16-rounded values)
          const float r0 = bf2f(o.x & 0xffffu), r1 = bf2f(o.x >> 16), r2 = bf2f(o.y & 0xffffu), r3 = bf2f(o.y >> 16);
          s1[mt][0] += r0; s1[mt][1] += r1; s1[mt][2] += r2; s1[mt][3] += r3;
          s2[mt][0] += r0 * r0; s2[mt][1] += r1 * r1; s2[mt][2] += r2 * r2; s2[mt][3] += r3 * r3;
        }
      }
    }
    TM(6);
  }
  if (a.stats_partial) {
    // reduce over the 16 voxel lanes of each g, then over the 4 waves through LDS
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);  // [wave][2][MT*16]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x1 = s1[mt][i], x2 = s2[mt][i];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
          x1 += __shfl_xor(x1, o, 64);
          x2 += __shfl_xor(x2, o, 64);
        }
        if (m == 0) {
          red[(wave * 2 + 0) * (MT * 16) + mt * 16 + 4 * g + i] = x1;
          red[(wave * 2 + 1) * (MT * 16) + mt * 16 + 4 * g + i] = x2;
        }
      }
    __syncthreads();
    float* dst = a.stats_partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (2 * MT * 16);
    for (int e = tid; e < 2 * MT * 16; e += 256)
      dst[e] = red[e] + red[2 * MT * 16 + e] + red[4 * MT * 16 + e] + red[6 * MT * 16 + e];
  }
}

template <int CK, int MT, bool WLDS>
int launch_fwd_w(const FwdArgs& a, int nchunks, hipStream_t st) {
  int gx = 512;
  while (gx > 8 && gx - 8 >= a.ntiles) gx -= 8;  // never narrower than the tile count: a second tile doubles a straggler's time
  if (a.ntiles < 8) gx = a.ntiles;
  constexpr int NSTEP = (27 * (CK / 8) + 3) / 4;
  const size_t hbytes = ((size_t)HVOX * rowb_fwd(CK) + 1023) / 1024 * 1024;
  const size_t smem = hbytes + (WLDS ? (size_t)NSTEP * MT * 1024 : 0);
  auto kern = conv3d_bf16_fwd_kernel<CK, MT, WLDS>;
  static SynOncePerDevice attr_done;
  if (auto once_ = attr_done.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  hipLaunchKernelGGL(kern, dim3(gx, nchunks, a.ksplit), dim3(256), smem, st, a);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

// folded variants (UPM 1: gridDim.z = 8 output parities; UPM 2: gridDim.z = K slices over the 8 x ncc_real chunks)
template <int CK, int MT, int UPM>
int launch_fwd_up(const FwdArgs& a, int nchunks, hipStream_t st) {
  int gx = UPM == 1 ? 64 : 512;  // UPM 1: 8 parities x 64 x chunks workgroups
  while (gx > 8 && gx - 8 >= a.ntiles) gx -= 8;  // never narrower than the tile count: a second tile doubles a straggler's time
  if (a.ntiles < 8) gx = a.ntiles;
  const size_t smem = ((size_t)HVOX * rowb_fwd(CK) + 1023) / 1024 * 1024;
  auto kern = conv3d_bf16_fwd_kernel<CK, MT, false, UPM>;
  static SynOncePerDevice attr_done;
  if (auto once_ = attr_done.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  hipLaunchKernelGGL(kern, dim3(gx, nchunks, UPM == 1 ? 8 : a.ksplit), dim3(256), smem, st, a);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}
template <int UPM>
int dispatch_fwd_up(const FwdArgs& a, const Bf16Plan& pl, hipStream_t st) {
#define SYN_UP(CKV, MTV) return launch_fwd_up<CKV, MTV, UPM>(a, pl.nchunks, st)
  if (pl.ck == 8) {
    if (pl.mt == 1) SYN_UP(8, 1); else if (pl.mt == 2) SYN_UP(8, 2); else if (pl.mt == 3) SYN_UP(8, 3); else SYN_UP(8, 4);
  } else if (pl.ck == 24) {
    if (pl.mt == 1) SYN_UP(24, 1); else if (pl.mt == 2) SYN_UP(24, 2); else if (pl.mt == 3) SYN_UP(24, 3); else SYN_UP(24, 4);
  } else {
    if (pl.mt == 1) SYN_UP(32, 1); else if (pl.mt == 2) SYN_UP(32, 2); else if (pl.mt == 3) SYN_UP(32, 3); else SYN_UP(32, 4);
  }
#undef SYN_UP
}

template <int CK, int MT>
int launch_fwd(const FwdArgs& a, int nchunks, hipStream_t st, int* wgs_out) {
  (void)wgs_out;
  // weights in LDS when there is a single input-channel chunk and two workgroups still fit a CU (<= 80 KB each)
  if constexpr (MT <= 2 && CK <= 24) {
    if (a.ncc == 1 && a.ksplit == 1) return launch_fwd_w<CK, MT, true>(a, nchunks, st);
  }
  return launch_fwd_w<CK, MT, false>(a, nchunks, st);
}

// split-K epilogue: fp32 partial sums [n4 x 4] -> + bias, activation, bf16
__global__ void bf16_epilogue_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                     const bf16_t* __restrict__ below, bf16_t* __restrict__ out, int64_t n4, int C4,
                                     int act, int ksplit, float alpha) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    float4 p = *reinterpret_cast<const float4*>(partial + i * 4);
    for (int k = 1; k < ksplit; ++k) {  // fixed order: deterministic
      const float4 q = *reinterpret_cast<const float4*>(partial + ((int64_t)k * n4 + i) * 4);
      p.x += q.x; p.y += q.y; p.z += q.z; p.w += q.w;
    }
    float v[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += bias ? bias[c + k] : 0.f;
    if (act == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = elu_f(v[k]);
    } else if (act == 2) {
      const u32x2 b = *reinterpret_cast<const u32x2*>(below + i * 4);
      v[0] *= elu_dy(bf2f(b.x & 0xffffu));
      v[1] *= elu_dy(bf2f(b.x >> 16));
      v[2] *= elu_dy(bf2f(b.y & 0xffffu));
      v[3] *= elu_dy(bf2f(b.y >> 16));
    } else if (act == 3) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : alpha * v[k];
    } else if (act == 4) {
      const u32x2 b = *reinterpret_cast<const u32x2*>(below + i * 4);
      v[0] *= bf2f(b.x & 0xffffu) > 0.f ? 1.f : alpha;
      v[1] *= bf2f(b.x >> 16) > 0.f ? 1.f : alpha;
      v[2] *= bf2f(b.y & 0xffffu) > 0.f ? 1.f : alpha;
      v[3] *= bf2f(b.y >> 16) > 0.f ? 1.f : alpha;
    } else if (act == 5) {
      const u32x2 b = *reinterpret_cast<const u32x2*>(below + i * 4);
      v[0] = elu_f(v[0] + bf2f(b.x & 0xffffu));
      v[1] = elu_f(v[1] + bf2f(b.x >> 16));
      v[2] = elu_f(v[2] + bf2f(b.y & 0xffffu));
      v[3] = elu_f(v[3] + bf2f(b.y >> 16));
    }
    u32x2 o;
    o.x = pack2(v[0], v[1]);
    o.y = pack2(v[2], v[3]);
    *reinterpret_cast<u32x2*>(out + i * 4) = o;
  }
}

// stride-2 'same' Conv3D of an even-sized volume (TensorFlow pads (0, 1): output o reads inputs 2o .. 2o+2) = the
// stride-1 'same' conv sampled at the ODD positions 2o + 1.  out[o] = in[2o + 1] (* LeakyReLU'(below[o]) when below is
// given: the masked forward pass of the gradient penalty); channels in 4-packs
__global__ void bf16_subsample_odd_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                          const bf16_t* __restrict__ below, int o0, int o1, int o2, int C4, float alpha) {
  const int64_t n4 = (int64_t)o0 * o1 * o2 * C4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    int64_t v = i / C4;
    const int x = (int)(v % o2);
    v /= o2;
    const int y = (int)(v % o1), z = (int)(v / o1);
    const int64_t src = (((int64_t)(2 * z + 1) * (2 * o1) + (2 * y + 1)) * (2 * o2) + (2 * x + 1)) * C4 + c4;
    u32x2 r = *reinterpret_cast<const u32x2*>(in + src * 4);
    if (below) {
      const u32x2 b = *reinterpret_cast<const u32x2*>(below + i * 4);
      const float f0 = bf2f(r.x & 0xffffu) * (bf2f(b.x & 0xffffu) > 0.f ? 1.f : alpha);
      const float f1 = bf2f(r.x >> 16) * (bf2f(b.x >> 16) > 0.f ? 1.f : alpha);
      const float f2 = bf2f(r.y & 0xffffu) * (bf2f(b.y & 0xffffu) > 0.f ? 1.f : alpha);
      const float f3 = bf2f(r.y >> 16) * (bf2f(b.y >> 16) > 0.f ? 1.f : alpha);
      r.x = pack2(f0, f1);
      r.y = pack2(f2, f3);
    }
    *reinterpret_cast<u32x2*>(out + i * 4) = r;
  }
}

// transpose of the above: out (full resolution, 2x) = in at the odd positions, zero elsewhere
__global__ void bf16_zero_insert_odd_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int o0, int o1, int o2,
                                            int C4) {
  const int64_t n4 = (int64_t)8 * o0 * o1 * o2 * C4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    int64_t v = i / C4;
    const int x = (int)(v % (2 * o2));
    v /= (2 * o2);
    const int y = (int)(v % (2 * o1)), z = (int)(v / (2 * o1));
    u32x2 r = {0u, 0u};
    if ((x & y & z) & 1) r = *reinterpret_cast<const u32x2*>(in + ((((int64_t)(z >> 1) * o1 + (y >> 1)) * o2 + (x >> 1)) * C4 + c4) * 4);
    *reinterpret_cast<u32x2*>(out + i * 4) = r;
  }
}

// batch statistics of a small bf16 tensor [nvox][C]: one workgroup per channel (split-K path only)
__global__ __launch_bounds__(256) void bf16_small_stats_kernel(const bf16_t* __restrict__ x, int64_t nvox, int C,
                                                               float* __restrict__ stats) {
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int64_t v = threadIdx.x; v < nvox; v += 256) {
    const double t = (double)bf2f(x[v * C + c]);
    s1 += t;
    s2 += t * t;
  }
  __shared__ double r1[256], r2[256];
  r1[threadIdx.x] = s1;
  r2[threadIdx.x] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      r1[threadIdx.x] += r1[threadIdx.x + o];
      r2[threadIdx.x] += r2[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double m = r1[0] / (double)nvox;
    double var = r2[0] / (double)nvox - m * m;
    stats[c] = (float)m;
    stats[C + c] = (float)(var < 0.0 ? 0.0 : var);
  }
}

// per-workgroup partials [nwg][2][W] (W = 16*MT columns per co-chunk) -> stats[mean C | var C] (double accumulation)
__global__ __launch_bounds__(64) void bf16_stats_finalize_kernel(const float* __restrict__ partial, int gx, int nchunks,
                                                                 int W, int C, float* __restrict__ stats, double inv_n) {
  const int c = blockIdx.x;  // one wave per channel
  const int chunk = c / W, col = c - chunk * W;
  double s1 = 0.0, s2 = 0.0;
  for (int w = threadIdx.x; w < gx; w += 64) {
    const float* p = partial + ((int64_t)chunk * gx + w) * (2 * W);
    s1 += (double)p[col];
    s2 += (double)p[W + col];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if (threadIdx.x == 0) {
    const double mean = s1 * inv_n;
    double var = s2 * inv_n - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[c] = (float)mean;
    stats[C + c] = (float)var;
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// gfx950 transpose read (ds_read_b64_tr_b16) through the compiler builtin: the register allocator places the two 64-bit
// halves of an MFMA operand in one aligned VGPR quad (inline asm needed 4 v_mov per operand), folds constant address parts
// into the 16-bit offset field and keeps its own s_waitcnt bookkeeping
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__device__ __forceinline__ bf16x8 tr_read8(const unsigned char* lds_base, uint32_t off_lo, uint32_t off_hi) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_base + off_lo));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_base + off_hi));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

struct WgArgs {
  const bf16_t* in;    // x [vox][Cin]
  const bf16_t* dout;  // dz [vox][Cout] (UP: on the 2x grid)
  float* dw;           // [27][cin_total][Cout] fp32, accumulated with atomics (UP: [8 parities][27 slots][cin_total][Cout])
  float* dbias;        // [Cout] or null
  int D0, D1, D2, Cin, Cout, cin_total, ci_off, ncc, nco, tiles1, tiles2, ntiles;
  int cin_valid;       // channels [0, cin_valid) of `in` have a row in dw (the zero-padded first layer: 2 of 8)
  int64_t det_stride;  // deterministic mode: dw / dbias are per-workgroup-column planes (common.h: DetRun), else 0
};

// NT: 16-column tiles of output channels per workgroup pass (<= 3)
// UP: weight gradient of the up-sampled channel range of a folded decoder conv: x = lo on the LOW-resolution grid (D0..D2),
// dz on the 2x grid; blockIdx.z = parity p: rows = the 8 taps (a, b, c) of the parity's 2x2x2 window (halo offset a + p per
// axis), dz is read on the sub-lattice 2 v + p; the partial goes to slot (a + p) of the 27-slot set of parity p in dwc
// (synthsr_conv3d_up_unpack folds the 8 sets back onto the 27 taps).
template <int CK, int NT, bool UP = false>
__global__ __launch_bounds__(512, (NT <= 2 ? 2 : 1)) void conv3d_bf16_wgrad_kernel(const WgArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int C8 = CK / 8, C4 = CK / 4, ROWB = rowb_for(CK);
  constexpr int NTAPS = UP ? 8 : 27;
  constexpr int NBLK = NTAPS * C4 + 1;              // (tap, quad) blocks + the constant-1 block (dbias)
  constexpr int NMT = (NBLK + 3) / 4;               // 16-row tiles
  constexpr int NW = 8, NTHR = 64 * NW;              // 8 waves: the row tiles' accumulators fit 128 VGPRs per wave
  constexpr int MPW = (NMT + NW - 1) / NW;          // row tiles per wave
  constexpr int DROWB = (NT & 1) ? NT * 32 : NT * 32 + 32;  // dz tile row stride: 32 B x odd (see rowb_for)
  constexpr int XBYTES = HVOX * ROWB;
  constexpr int NPIECE = HVOX * C8, NLD = (NPIECE + NTHR - 1) / NTHR;
  constexpr int NDP = TZ * TY * TX * NT * 2, NDL = (NDP + NTHR - 1) / NTHR;  // 16-byte pieces of the dz tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15, lrow = li >> 2, lq = li & 3;
  const int cc = blockIdx.y % a.ncc, oc = blockIdx.y / a.ncc;
  const TileWalk walk = tile_walk(a.ntiles);
  const int tiles0 = a.ntiles / (a.tiles1 * a.tiles2);
  const int D0 = a.D0, D1 = a.D1, D2 = a.D2, Cin = a.Cin, Cout = a.Cout;
  unsigned char* ldz = lds + XBYTES;

  // A addresses: lane i = (row lrow -> voxel 8g + lrow [+4], chunk lq -> block 4*mtile + lq)
  int aoff[MPW];
#pragma unroll
  for (int q = 0; q < MPW; ++q) {
    const int blk = (wave * MPW + q) * 4 + lq;
    int off;
    if (blk < NTAPS * C4) {
      const int tap = blk / C4, quad = blk - tap * C4;
      const int par = UP ? (int)blockIdx.z : 0;
      const int hz = UP ? (tap >> 2) + ((par >> 2) & 1) : tap / 9, hy = UP ? ((tap >> 1) & 1) + ((par >> 1) & 1) : (tap / 3) % 3,
                hx = UP ? (tap & 1) + (par & 1) : tap % 3;
      off = ((hz * HY + hy) * HX + hx) * ROWB + quad * 8;
    } else if (blk == NTAPS * C4) {
      off = ((HY + 1) * HX + 1) * ROWB + CK * 2;      // centre tap, pad channels CK..CK+3 = (1, 0, 0, 0)
    } else {
      off = ((HY + 1) * HX + 1) * ROWB + CK * 2 + 8;  // unused slots: pad channels CK+4..CK+7 = 0
    }
    aoff[q] = off;
  }
  // K index 8g + j of a 32-voxel K-step (2 x-rows) <-> voxel (row g >> 1, x = 8 (j >> 2) + 4 (g & 1) + (j & 3)): any
  // bijection works as long as A and B agree, and with this one the 32 lanes an LDS cycle serves (g and g ^ 1) read 8
  // CONSECUTIVE x-voxels -- with 96-byte rows their 32-byte pieces fall into 8 different bank groups (no conflicts when the
  // 4 blocks of a row tile share a tap; with voxel 8g + j, x and x + 8 always met in the same banks)
  const int vx = 4 * (g & 1) + lrow, vr = g >> 1;
  const uint32_t abase = (uint32_t)((vr * HX + vx) * ROWB);            // + aoff[q]; K-step rows / x + 4 as immediates
  const uint32_t bbase = (uint32_t)XBYTES + (uint32_t)((vr * TX + vx) * DROWB + lq * 8);

  // pad channels of every halo row (never overwritten by the staging)
  for (int v = tid; v < HVOX; v += NTHR) {
    u32x4 one = {0x00003f80u, 0u, 0u, 0u};  // bf16 1.0 in channel CK
    *reinterpret_cast<u32x4*>(lds + v * ROWB + CK * 2) = one;
  }

  int prel[NLD], plds[NLD];
  uint32_t pmask[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int j = tid + NTHR * i;
    const int v = j / C8, c8 = j - v * C8;
    const int hz = v / (HY * HX), r = v - hz * (HY * HX), hy = r / HX, hx = r - hy * HX;
    prel[i] = ((hz * D1 + hy) * D2 + hx) * Cin * 2 + c8 * 16;
    plds[i] = v * ROWB + c8 * 16;
    pmask[i] = j < NPIECE ? ((1u << hz) | (1u << (6 + hy)) | (1u << (12 + hx))) : 0xFFFFFFFFu;  // hx <= 17 -> bit 29
  }
  // dz pieces: piece j -> tile voxel j / (2 NT), 16-byte group j % (2 NT) (8 channels)
  int drel[NDL], dlds[NDL], dvz[NDL], dvy[NDL], dvx[NDL];
  bool dcok[NDL];
#pragma unroll
  for (int i = 0; i < NDL; ++i) {
    const int j = tid + NTHR * i;
    const int v = j / (2 * NT), c8 = j - v * (2 * NT);
    dvz[i] = v / (TY * TX);
    dvy[i] = (v / TX) % TY;
    dvx[i] = v % TX;
    const int co = oc * NT * 16 + c8 * 8;
    dcok[i] = j < NDP && co < Cout;                     // Cout is a multiple of 8
    drel[i] = UP ? ((2 * dvz[i] * (2 * D1) + 2 * dvy[i]) * (2 * D2) + 2 * dvx[i]) * Cout * 2 + co * 2
                 : ((dvz[i] * D1 + dvy[i]) * D2 + dvx[i]) * Cout * 2 + co * 2;
    dlds[i] = v * DROWB + c8 * 16;
  }
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(a.in), 0, (int)((int64_t)D0 * D1 * D2 * Cin * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(a.dout), 0, (int)((int64_t)(UP ? 8 : 1) * D0 * D1 * D2 * Cout * 2), 0x00020000);
  u32x4 stg[NLD], dst[NDL];
  auto tile_origin = [&](int t, int& z0, int& y0, int& x0) { tile_decode(t, tiles0, a.tiles1, a.tiles2, z0, y0, x0); };
  auto load_tile = [&](int t) {
    int z0, y0, x0;
    tile_origin(t, z0, y0, x0);
    uint32_t bad = 0x80000000u;
#pragma unroll
    for (int h = 0; h < HZ; ++h) bad |= ((unsigned)(z0 - 1 + h) >= (unsigned)D0) ? (1u << h) : 0u;
#pragma unroll
    for (int h = 0; h < HY; ++h) bad |= ((unsigned)(y0 - 1 + h) >= (unsigned)D1) ? (1u << (6 + h)) : 0u;
#pragma unroll
    for (int h = 0; h < HX; ++h) bad |= ((unsigned)(x0 - 1 + h) >= (unsigned)D2) ? (1u << (12 + h)) : 0u;
    const int base = ((((z0 - 1) * D1 + (y0 - 1)) * D2 + (x0 - 1)) * Cin + cc * CK) * 2;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const uint32_t vo = (pmask[i] & bad) ? OOB : (uint32_t)(prel[i] + base);
      stg[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)vo, 0, 0);
    }
    const int dpar = UP ? (int)blockIdx.z : 0;
    const int dbase = UP ? (((2 * z0 + ((dpar >> 2) & 1)) * (2 * D1) + (2 * y0 + ((dpar >> 1) & 1))) * (2 * D2) +
                            (2 * x0 + (dpar & 1))) * Cout * 2
                         : ((z0 * D1 + y0) * D2 + x0) * Cout * 2;
#pragma unroll
    for (int i = 0; i < NDL; ++i) {
      const bool ok = dcok[i] && z0 + dvz[i] < D0 && y0 + dvy[i] < D1 && x0 + dvx[i] < D2;
      dst[i] = __builtin_amdgcn_raw_buffer_load_b128(rdo, ok ? drel[i] + dbase : (int)OOB, 0, 0);
    }
  };

  f32x4 acc[MPW][NT];
#pragma unroll
  for (int q = 0; q < MPW; ++q)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[q][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (walk.pos < walk.end) load_tile(walk.pos);
  for (int t = walk.pos; t < walk.end; t += walk.stride) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      if (i < NLD - 1 || tid + NTHR * i < NPIECE) *reinterpret_cast<u32x4*>(lds + plds[i]) = stg[i];
#pragma unroll
    for (int i = 0; i < NDL; ++i)
      if (i < NDL - 1 || tid + NTHR * i < NDP) *reinterpret_cast<u32x4*>(ldz + dlds[i]) = dst[i];
    __syncthreads();
    if (t + walk.stride < walk.end) load_tile(t + walk.stride);
    // 8 K-steps of 32 voxels = 2 x-rows each: rows (z, y) = (ks >> 1, 2 (ks & 1) + vr)
    sfor<0, 8>([&](auto KS) {
      constexpr int ks = decltype(KS)::value;
      constexpr int z = ks >> 1, yb = 2 * (ks & 1);
      constexpr int arow = ((z * HY + yb) * HX) * ROWB;     // halo row of voxel (z, yb, 0) with tap (0, 0, 0)
      constexpr int brow = ((z * TY + yb) * TX) * DROWB;
      // all transpose reads of the K-step first (B: 2 per column tile, A: 2 per row tile), then the MFMAs
      bf16x8 bfr[NT], afr[MPW];
#pragma unroll
      for (int n = 0; n < NT; ++n) bfr[n] = tr_read8(lds, bbase + brow + n * 32, bbase + brow + n * 32 + 8 * DROWB);
#pragma unroll
      for (int q = 0; q < MPW; ++q) afr[q] = tr_read8(lds, abase + aoff[q] + arow, abase + aoff[q] + arow + 8 * ROWB);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < MPW; ++q) {
        if (wave * MPW + q >= NMT) continue;  // wave-uniform
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[q][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[q], bfr[n], acc[q][n], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  // ---- flush: lane (n = li -> co, rows 4g + i -> block 4*mtile + g, element i); every address belongs to one lane of the
  // workgroup, deterministic mode gives each workgroup column its own plane (a.det_stride)
  const size_t detoff = (size_t)blockIdx.x * a.det_stride;
  const int fpar = UP ? (int)blockIdx.z : 0;
  float* dwp = a.dw + detoff + (UP ? (int64_t)fpar * 27 * a.cin_total * Cout : (int64_t)0);
#pragma unroll
  for (int q = 0; q < MPW; ++q) {
    const int mtile = wave * MPW + q;
    if (mtile >= NMT) continue;
    const int blk = mtile * 4 + g;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int co = (oc * NT + n) * 16 + li;
      if (co >= Cout) continue;
      if (blk < NTAPS * C4) {
        const int tap8 = blk / C4, quad = blk - tap8 * C4;
        // UP: tap (a, b, c) of parity p is slot (a + pz, b + py, c + px) of that parity's 27-slot set
        const int tap = UP ? (((tap8 >> 2) + ((fpar >> 2) & 1)) * 3 + (((tap8 >> 1) & 1) + ((fpar >> 1) & 1))) * 3 +
                                 ((tap8 & 1) + (fpar & 1))
                           : tap8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ci = cc * CK + quad * 4 + i;
          if (ci < a.cin_valid) atomicAdd(dwp + ((int64_t)tap * a.cin_total + a.ci_off + ci) * Cout + co, acc[q][n][i]);
        }
      } else if (blk == NTAPS * C4 && a.dbias && cc == 0) {
        atomicAdd(a.dbias + detoff + co, acc[q][n][0]);
      }
    }
  }
}

template <int CK, int NT, bool UP = false>
int launch_wgrad(const WgArgs& a0, hipStream_t st) {
  WgArgs a = a0;
  int gx = 512 / (a.ncc * a.nco * (UP ? 8 : 1));
  gx = (gx / 8) * 8;
  if (gx < 8) gx = 8;
  while (gx > 8 && gx - 8 >= a.ntiles) gx -= 8;  // never narrower than the tile count: a second tile doubles a straggler's time
  if (a.ntiles < 8) gx = a.ntiles;
  const size_t smem = (size_t)HVOX * rowb_for(CK) + (size_t)TZ * TY * TX * ((NT & 1) ? NT * 32 : NT * 32 + 32);
  auto kern = conv3d_bf16_wgrad_kernel<CK, NT, UP>;
  static SynOncePerDevice attr_done;
  if (auto once_ = attr_done.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  DetRun det;
  if (const int rc_ = syn_det_prepare(&det, &a.dw, &a.dbias, (int64_t)(UP ? 8 : 1) * 27 * a.cin_total * a.Cout, a.Cout, gx, st)) return rc_;
  a.det_stride = det.stride;
  hipLaunchKernelGGL(kern, dim3(gx, a.ncc * a.nco, UP ? 8 : 1), dim3(512), smem, st, a);
  if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
  return syn_det_finish(&det, st);
}

// fp32 [n][Cs] -> bf16 [n][Cd] (Cd >= Cs, zero fill): the generator's image -> first-layer input (Cin 2 -> 8)
__global__ void f32_to_bf16_pad_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n, int Cs, int Cd) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n * Cd; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / Cd;
    const int c = (int)(i - v * Cd);
    dst[i] = (bf16_t)(c < Cs ? f2bf(src[v * Cs + c]) : 0u);
  }
}

}  // namespace

extern "C" {

int64_t synthsr_conv3d_bf16_pack_ex(const float* w, void* packed, int Cin_total, int ci_off, int Cin, int Cout, int mode,
                                    int parity, synthsr_stream_t stream) {
  if (Cin < 1 || Cout < 1 || ci_off < 0 || ci_off + Cin > Cin_total || (mode != 0 && mode != 1) || parity < -1 || parity > 7)
    return SYNTHSR_EINVAL;
  // the tensor a layer reads has its channel count padded to a multiple of 8 (the first layer's 2 -> 8): pad channels
  // get zero weights.  The tensor it produces has its channel count padded the same way when CoutE is not a multiple
  // of 4 (data gradient of a 1- or 2-channel first layer): the extra rows are zero, the tile count is the same.
  const int CinE = ((mode ? Cout : Cin) + 7) / 8 * 8, CoutE = mode ? Cin : Cout;
  const Bf16Plan pl = plan_bf16(CinE, CoutE, parity < 0 ? 27 : 8);
  const int64_t total = pl.count();
  if (total >= (1ll << 31)) return SYNTHSR_EINVAL;
  if (!packed) return total;
  if (!w) return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(pack_bf16_kernel, dim3(syn_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)packed,
                     Cin_total, ci_off, Cin, Cout, mode, pl.ck, pl.ncc, pl.mt, pl.nsteps, parity, total);
  return hipGetLastError() == hipSuccess ? total : (int64_t)SYNTHSR_ELAUNCH;
}

int64_t synthsr_conv3d_bf16_pack(const float* w, void* packed, int Cin_total, int ci_off, int Cin, int Cout, int mode,
                                 synthsr_stream_t stream) {
  return synthsr_conv3d_bf16_pack_ex(w, packed, Cin_total, ci_off, Cin, Cout, mode, -1, stream);
}

// job table row of synthsr_conv3d_bf16_pack_all for one weight set (host side; fields 0, 1 = w_off, dst_off are the caller's);
// parity -1: the plain 27-tap set, 0..7: one parity set of the nearest-upsample folding
int synthsr_conv3d_bf16_pack_job(int Cin_total, int ci_off, int Cin, int Cout, int mode, int parity, int64_t job[13]) {
  if (!job || Cin < 1 || Cout < 1 || ci_off < 0 || ci_off + Cin > Cin_total || (mode != 0 && mode != 1) || parity < -1 ||
      parity > 7)
    return SYNTHSR_EINVAL;
  const int CinE = ((mode ? Cout : Cin) + 7) / 8 * 8, CoutE = mode ? Cin : Cout;
  const Bf16Plan pl = plan_bf16(CinE, CoutE, parity < 0 ? 27 : 8);
  if (pl.count() >= (1ll << 31)) return SYNTHSR_EINVAL;
  job[2] = pl.count();
  job[3] = Cin_total;
  job[4] = ci_off;
  job[5] = Cin;
  job[6] = Cout;
  job[7] = mode;
  job[8] = pl.ck;
  job[9] = pl.ncc;
  job[10] = pl.mt;
  job[11] = pl.nsteps;
  job[12] = parity;
  return SYNTHSR_OK;
}

int synthsr_conv3d_bf16_pack_all(const float* params, void* packed, const int64_t* jobs_dev, int njobs,
                                 synthsr_stream_t stream) {
  if (!params || !packed || !jobs_dev || njobs < 1) return SYNTHSR_EINVAL;
  hipLaunchKernelGGL(pack_bf16_all_kernel, dim3(128, njobs), dim3(256), 0, (hipStream_t)stream, params, (bf16_t*)packed,
                     jobs_dev);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

int synthsr_conv3d_bf16_fwd_ex(const void* in, const void* wp, const float* bias, void* out, const int shape[3], int Cin,
                               int Cout, int act, float alpha, const void* below, float* stats, float* scratch,
                               int64_t scratch_floats, synthsr_stream_t stream) {
  if (!in || !wp || !out || !shape || Cin % 8 != 0 || Cout % 4 != 0 || act < 0 || act > 5 ||
      ((act == 2 || act == 4 || act == 5) && !below) || (act == 5 && stats))
    return SYNTHSR_EINVAL;
  const int64_t vox = (int64_t)shape[0] * shape[1] * shape[2];
  if (vox * Cin * 2 >= (1ll << 31) || vox * Cout * 2 >= (1ll << 31)) return SYNTHSR_EINVAL;
  const Bf16Plan pl = plan_bf16(Cin, Cout);
  FwdArgs a;
  a.in = (const bf16_t*)in;
  a.wp = (const bf16_t*)wp;
  a.bias = bias;
  a.out = (bf16_t*)out;
  a.below = (const bf16_t*)below;
  a.D0 = shape[0];
  a.D1 = shape[1];
  a.D2 = shape[2];
  a.Cin = Cin;
  a.Cout = Cout;
  a.ncc = pl.ncc;
  a.tiles1 = (shape[1] + TY - 1) / TY;
  a.tiles2 = (shape[2] + TX - 1) / TX;
  a.ntiles = ((shape[0] + TZ - 1) / TZ) * a.tiles1 * a.tiles2;
  a.act = act;
  a.alpha = alpha;
  a.stats_partial = nullptr;
  a.partial = nullptr;
  a.ksplit = 1;
  a.ncc_real = pl.ncc;
  int gx = 512;
  while (gx > 8 && gx - 8 >= a.ntiles) gx -= 8;  // never narrower than the tile count: a second tile doubles a straggler's time
  if (a.ntiles < 8) gx = a.ntiles;
  const int W = 16 * pl.mt;
  hipStream_t st = (hipStream_t)stream;
  // small deep levels (20^3, 10^3): a handful of tiles cannot fill 256 CUs and each workgroup would walk up to 18 channel
  // chunks one after the other -> split the chunks over gridDim.z, accumulate in fp32, finish in a second tiny kernel
  const int wgs_plain = gx * pl.nchunks;
  int ks = std::min(pl.ncc, (512 + wgs_plain - 1) / std::max(wgs_plain, 1));
  if (wgs_plain < 256 && ks >= 2 && scratch && scratch_floats >= (int64_t)ks * vox * Cout) {
    a.ksplit = ks;
    a.partial = scratch;
  } else if (stats) {
    if (!scratch || scratch_floats < (int64_t)gx * pl.nchunks * 2 * W) return SYNTHSR_EINVAL;
    a.stats_partial = scratch;
  }
  int rc = SYNTHSR_EINVAL, wgs = 0;
#define SYN_FWD(CKV, MTV) rc = launch_fwd<CKV, MTV>(a, pl.nchunks, st, &wgs)
  if (pl.ck == 8) {
    if (pl.mt == 1) SYN_FWD(8, 1); else if (pl.mt == 2) SYN_FWD(8, 2); else if (pl.mt == 3) SYN_FWD(8, 3); else SYN_FWD(8, 4);
  } else if (pl.ck == 24) {
    if (pl.mt == 1) SYN_FWD(24, 1); else if (pl.mt == 2) SYN_FWD(24, 2); else if (pl.mt == 3) SYN_FWD(24, 3); else SYN_FWD(24, 4);
  } else {
    if (pl.mt == 1) SYN_FWD(32, 1); else if (pl.mt == 2) SYN_FWD(32, 2); else if (pl.mt == 3) SYN_FWD(32, 3); else SYN_FWD(32, 4);
  }
#undef SYN_FWD
  if (rc != SYNTHSR_OK) return rc;
  if (a.partial) {
    const int64_t n4 = vox * (Cout / 4);
    hipLaunchKernelGGL(bf16_epilogue_kernel, dim3(syn_grid(n4, 256)), dim3(256), 0, st, scratch, bias, (const bf16_t*)below,
                       (bf16_t*)out, n4, Cout / 4, act, a.ksplit, a.alpha);
    if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
    if (stats) {
      hipLaunchKernelGGL(bf16_small_stats_kernel, dim3(Cout), dim3(256), 0, st, (const bf16_t*)out, vox, Cout, stats);
      if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
    }
    return SYNTHSR_OK;
  }
  if (stats) {
    hipLaunchKernelGGL(bf16_stats_finalize_kernel, dim3(Cout), dim3(64), 0, st, scratch, gx, pl.nchunks, W, Cout, stats,
                       1.0 / (double)vox);
    if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
  }
  return SYNTHSR_OK;
}

int synthsr_conv3d_bf16_fwd(const void* in, const void* wp, const float* bias, void* out, const int shape[3], int Cin,
                            int Cout, int act, const void* below, float* stats, float* scratch, int64_t scratch_floats,
                            synthsr_stream_t stream) {
  if (act > 2) return SYNTHSR_EINVAL;
  return synthsr_conv3d_bf16_fwd_ex(in, wp, bias, out, shape, Cin, Cout, act, 0.f, below, stats, scratch, scratch_floats,
                                    stream);
}

int synthsr_bf16_subsample_odd(const void* in, void* out, const void* below, const int out_shape[3], int C, float alpha,
                               synthsr_stream_t stream) {
  if (!in || !out || !out_shape || C < 4 || C % 4 != 0 || out_shape[0] < 1 || out_shape[1] < 1 || out_shape[2] < 1)
    return SYNTHSR_EINVAL;
  const int64_t n4 = (int64_t)out_shape[0] * out_shape[1] * out_shape[2] * (C / 4);
  hipLaunchKernelGGL(bf16_subsample_odd_kernel, dim3(syn_grid(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in,
                     (bf16_t*)out, (const bf16_t*)below, out_shape[0], out_shape[1], out_shape[2], C / 4, alpha);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

int synthsr_bf16_zero_insert_odd(const void* in, void* out, const int in_shape[3], int C, synthsr_stream_t stream) {
  if (!in || !out || !in_shape || C < 4 || C % 4 != 0 || in_shape[0] < 1 || in_shape[1] < 1 || in_shape[2] < 1)
    return SYNTHSR_EINVAL;
  const int64_t n4 = (int64_t)8 * in_shape[0] * in_shape[1] * in_shape[2] * (C / 4);
  hipLaunchKernelGGL(bf16_zero_insert_odd_kernel, dim3(syn_grid(n4, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)in, (bf16_t*)out, in_shape[0], in_shape[1], in_shape[2], C / 4);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

int64_t synthsr_conv3d_bf16_stats_scratch(const int shape[3], int Cin, int Cout) {
  if (!shape) return SYNTHSR_EINVAL;
  const Bf16Plan pl = plan_bf16(Cin, Cout);
  const int64_t stats = (int64_t)512 * pl.nchunks * 2 * 16 * pl.mt;
  const int tiles = ((shape[0] + TZ - 1) / TZ) * ((shape[1] + TY - 1) / TY) * ((shape[2] + TX - 1) / TX);
  const int wgs = std::min(tiles, 512) * pl.nchunks;
  const int64_t ks = std::min(pl.ncc, (512 + wgs - 1) / std::max(wgs, 1));
  const int64_t splitk = wgs < 256 ? ks * shape[0] * shape[1] * shape[2] * Cout : 0;  // fp32 partial planes
  return stats > splitk ? stats : splitk;
}

static int bf16_wgrad_common(const void* in, const void* dout, float* dw, float* dbias, const int shape[3], int dw_cin_total,
                             int ci_off, int Cin, int cin_valid, int Cout, bool up, hipStream_t st) {
  const int64_t vox = (int64_t)shape[0] * shape[1] * shape[2];
  if (vox * Cin * 2 >= (1ll << 31) || (up ? 8 : 1) * vox * Cout * 2 >= (1ll << 31)) return SYNTHSR_EINVAL;
  WgArgs a;
  a.in = (const bf16_t*)in;
  a.dout = (const bf16_t*)dout;
  a.dw = dw;
  a.dbias = dbias;
  a.D0 = shape[0];
  a.D1 = shape[1];
  a.D2 = shape[2];
  a.Cin = Cin;
  a.Cout = Cout;
  a.cin_total = dw_cin_total;
  a.ci_off = ci_off;
  a.cin_valid = cin_valid;
  a.det_stride = 0;
  const int ck = (Cin % 32 == 0) ? 32 : ((Cin % 24 == 0) ? 24 : 8);
  a.ncc = (Cin + ck - 1) / ck;
  const int nt_all = (Cout + 15) / 16;
  a.nco = (nt_all + 2) / 3;
  const int nt = (nt_all + a.nco - 1) / a.nco;
  a.tiles1 = (shape[1] + TY - 1) / TY;
  a.tiles2 = (shape[2] + TX - 1) / TX;
  a.ntiles = ((shape[0] + TZ - 1) / TZ) * a.tiles1 * a.tiles2;
#define SYN_WG(CKV, UPV) \
  (nt == 1 ? launch_wgrad<CKV, 1, UPV>(a, st) : (nt == 2 ? launch_wgrad<CKV, 2, UPV>(a, st) : launch_wgrad<CKV, 3, UPV>(a, st)))
  if (up) {
    if (ck == 8) return SYN_WG(8, true);
    if (ck == 24) return SYN_WG(24, true);
    return SYN_WG(32, true);
  }
  if (ck == 8) return SYN_WG(8, false);
  if (ck == 24) return SYN_WG(24, false);
  return SYN_WG(32, false);
#undef SYN_WG
}

int synthsr_conv3d_bf16_wgrad(const void* in, const void* dout, float* dw, float* dbias, const int shape[3], int Cin_total,
                              int ci_off, int Cin, int Cout, synthsr_stream_t stream) {
  // `in` has Cin channels (a multiple of 8), dw covers its first Cin_total <= Cin channels (zero-padded first layer)
  if (!in || !dout || !dw || !shape || Cin % 8 != 0 || Cout % 8 != 0 || ci_off != 0 || Cin_total > Cin || Cin_total < 1)
    return SYNTHSR_EINVAL;
  return bf16_wgrad_common(in, dout, dw, dbias, shape, Cin_total, 0, Cin, Cin_total, Cout, false, (hipStream_t)stream);
}

int synthsr_conv3d_bf16_wgrad_part(const void* in, const void* dout, float* dw, float* dbias, const int shape[3],
                                   int Cin_total, int ci_off, int Cin, int Cout, synthsr_stream_t stream) {
  // the Cin channels of `in` are the input-channel range [ci_off, ci_off + Cin) of a layer with Cin_total input channels
  if (!in || !dout || !dw || !shape || Cin % 8 != 0 || Cout % 8 != 0 || ci_off < 0 || ci_off + Cin > Cin_total)
    return SYNTHSR_EINVAL;
  return bf16_wgrad_common(in, dout, dw, dbias, shape, Cin_total, ci_off, Cin, Cin, Cout, false, (hipStream_t)stream);
}

int synthsr_conv3d_bf16_up_wgrad(const void* lo, const void* dout, float* dwc, const int lo_shape[3], int Cl, int Cout,
                                 synthsr_stream_t stream) {
  if (!lo || !dout || !dwc || !lo_shape || Cl % 8 != 0 || Cout % 8 != 0 || Cl < 8 || Cout < 8) return SYNTHSR_EINVAL;
  return bf16_wgrad_common(lo, dout, dwc, nullptr, lo_shape, Cl, 0, Cl, Cl, Cout, true, (hipStream_t)stream);
}

static int bf16_up_args(FwdArgs& a, const void* in, const void* wp, void* out, const int lo_shape[3], int Cin, int Cout) {
  a.in = (const bf16_t*)in;
  a.wp = (const bf16_t*)wp;
  a.bias = nullptr;
  a.out = (bf16_t*)out;
  a.below = nullptr;
  a.D0 = lo_shape[0];
  a.D1 = lo_shape[1];
  a.D2 = lo_shape[2];
  a.Cin = Cin;
  a.Cout = Cout;
  a.tiles1 = (lo_shape[1] + TY - 1) / TY;
  a.tiles2 = (lo_shape[2] + TX - 1) / TX;
  a.ntiles = ((lo_shape[0] + TZ - 1) / TZ) * a.tiles1 * a.tiles2;
  a.act = 0;
  a.alpha = 0.f;
  a.stats_partial = nullptr;
  a.partial = nullptr;
  a.ksplit = 1;
  return SYNTHSR_OK;
}

int synthsr_conv3d_bf16_up_fwd(const void* lo, const void* wpacked8, void* out, const int lo_shape[3], int Cl, int Cout,
                               synthsr_stream_t stream) {
  if (!lo || !wpacked8 || !out || !lo_shape || Cl % 8 != 0 || Cout % 4 != 0 || Cl < 8 || Cout < 4) return SYNTHSR_EINVAL;
  const int64_t vox = (int64_t)lo_shape[0] * lo_shape[1] * lo_shape[2];
  if (vox * Cl * 2 >= (1ll << 31) || 8 * vox * Cout * 2 >= (1ll << 31)) return SYNTHSR_EINVAL;
  const Bf16Plan pl = plan_bf16(Cl, Cout, 8);
  FwdArgs a;
  bf16_up_args(a, lo, wpacked8, out, lo_shape, Cl, Cout);
  a.ncc = pl.ncc;
  a.ncc_real = pl.ncc;
  return dispatch_fwd_up<1>(a, pl, (hipStream_t)stream);
}

int synthsr_conv3d_bf16_up_dgrad(const void* dout, const void* wpacked8, void* dlo, const int lo_shape[3], int Cl, int Cout,
                                 float* scratch, int64_t scratch_floats, synthsr_stream_t stream) {
  if (!dout || !wpacked8 || !dlo || !lo_shape || Cl % 4 != 0 || Cout % 8 != 0 || Cl < 4 || Cout < 8) return SYNTHSR_EINVAL;
  const int64_t vox = (int64_t)lo_shape[0] * lo_shape[1] * lo_shape[2];
  if (8 * vox * Cout * 2 >= (1ll << 31) || vox * Cl * 2 >= (1ll << 31)) return SYNTHSR_EINVAL;
  const Bf16Plan pl = plan_bf16(Cout, Cl, 8);  // K = dz channels (Cout of the layer), output = lo channels
  FwdArgs a;
  bf16_up_args(a, dout, wpacked8, dlo, lo_shape, Cout, Cl);
  a.ncc_real = pl.ncc;
  a.ncc = 8 * pl.ncc;
  hipStream_t st = (hipStream_t)stream;
  // small deep levels: split the 8 x ncc K chunks over gridDim.z, fp32 partial planes + the split-K epilogue
  int gx = 512;
  while (gx > 8 && gx - 8 >= a.ntiles) gx -= 8;  // never narrower than the tile count: a second tile doubles a straggler's time
  if (a.ntiles < 8) gx = a.ntiles;
  const int wgs_plain = gx * pl.nchunks;
  const int ks = std::min(a.ncc, (512 + wgs_plain - 1) / std::max(wgs_plain, 1));
  if (wgs_plain < 256 && ks >= 2 && scratch && scratch_floats >= (int64_t)ks * vox * Cl) {
    a.ksplit = ks;
    a.partial = scratch;
  }
  const int rc = dispatch_fwd_up<2>(a, pl, st);
  if (rc != SYNTHSR_OK) return rc;
  if (a.partial) {
    const int64_t n4 = vox * (Cl / 4);
    hipLaunchKernelGGL(bf16_epilogue_kernel, dim3(syn_grid(n4, 256)), dim3(256), 0, st, scratch, (const float*)nullptr,
                       (const bf16_t*)nullptr, (bf16_t*)dlo, n4, Cl / 4, 0, a.ksplit, 0.f);
    if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
  }
  return SYNTHSR_OK;
}

int synthsr_f32_to_bf16_pad(const float* src, void* dst, int64_t n, int Cs, int Cd, synthsr_stream_t stream) {
  if (!src || !dst || n < 0 || Cs < 1 || Cd < Cs) return SYNTHSR_EINVAL;
  if (n == 0) return SYNTHSR_OK;
  hipLaunchKernelGGL(f32_to_bf16_pad_kernel, dim3(syn_grid(n * Cd, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (bf16_t*)dst, n, Cs, Cd);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

}  // extern "C"
