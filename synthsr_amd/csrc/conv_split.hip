// 3x3x3 'same' Conv3D on NDHWC **fp32** activations / weights evaluated on the bf16 matrix cores by operand splitting
// (gfx950, v_mfma_f32_16x16x32_bf16, fp32 accumulation).  SynthSR's U-Net is fp32 Keras (ext/neuron/models.py:256-498,
// SynthSR/training.py:330-341); gfx950 multiplies bf16 16x faster than fp32 (2.5 PFLOP/s vs 157 TFLOP/s dense), so every
// fp32 operand is written as the exact sum of three bf16 numbers
//       a = a0 + a1 + a2,   a0 = bf16(a), a1 = bf16(a - a0), a2 = a - a0 - a1      (round to nearest even; 3 x 8 significand
//                                                                                  bits cover the 24 of fp32: a2 is exact)
// and a product a b is accumulated as the six partial products a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0, each EXACT in the
// matrix core's fp32 accumulator (8 x 8 significand bits).  The three products left out are bounded by
// |a1 b2 + a2 b1 + a2 b2| <= (2 . 2^-8 2^-16 + 2^-32) |a b| < 2^-23 |a b| (|a1| <= 2^-8 |a|, |a2| <= 2^-16 |a|); over random
// operands the omission measures 2^-24.2 |a b| at most and 2^-27.4 rms (tests/test_split_arithmetic_cpu.py) -- no more than the
// rounding an fp32 multiply-add commits on the product (2^-24), far below the rounding of the 648-term accumulation that
// follows, so the result is as close to the exact convolution as the fp32-MFMA kernels of conv3d.hip are
// (tests/test_split_gpu.py measures both against float64; all nine products would be exact and cost 1.5x the MFMAs).  Six bf16 MFMAs per fp32 MFMA's worth of work = 2.6x the fp32
// matrix rate; activations stay fp32 in HBM (the split happens on the way into LDS), weights are split when they are packed.
//
//   forward / data gradient:  D[co][voxel] += sum over the 6 (i, j) of  A_i[co][(tap, ci)] * B_j[(tap, ci)][voxel]
//       geometry of conv_bf16.hip's forward kernel: M = output channels (A = packed weight fragments, streamed from L2),
//       N = 16 voxels of an x-row, K = 4 taps x 8 channels per MFMA; 4x4x16-voxel tiles, 4 waves (wave = z plane),
//       persistent workgroups walking the tiles XCD-contiguously.  Input channels go through LDS 8 at a time as three bf16
//       planes [piece][halo voxel][8 ch] (3 x 10 KB), double buffered: while the 6 x 4 x MT MFMAs of a K step run on one
//       buffer, the next 8-channel chunk is converted and written to the other -- one barrier per chunk.
#include "common.h"
#include <algorithm>
#include <type_traits>
#include <utility>

#ifndef SYN_ABL  // ablation builds of the interleaved forward kernel (tools/split_ablate.sh): 1 no halo loads, 2 no conversion,
#define SYN_ABL 0  // 4 no LDS image stores, 8 weights loaded once, 16 activation fragments loaded once per chunk, 32 no stores, 64 no MFMAs
#endif
#ifdef SYN_SPLIT_TIMING  // per-phase shader-clock stamps of wave 0 of workgroups 0 and 300 (tools/split_phase_timing.py)
static __device__ long long* g_tm = nullptr;
extern "C" int synthsr_split_timing_buffer(long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_tm), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#define TM(i) do { if (g_tm && (blockIdx.x == 0 || blockIdx.x == 300) && blockIdx.y == 0 && tid == 0 && tix < 120) g_tm[((blockIdx.x ? 1 : 0) * 120 + tix) * 8 + (i)] = clock64(); } while (0)
// fwd2: sequential stamps of thread 0 of workgroups (7, 0) and (20, 1): T2(k) writes slot n2++ = (k, clock)
#define T2(k) do { if (g_tm && threadIdx.x == 0 && ((blockIdx.x == 7 && blockIdx.y == 0) || (blockIdx.x == 20 && blockIdx.y == 1)) && n2 < 400) { g_tm[(blockIdx.y * 400 + n2) * 2] = (k); g_tm[(blockIdx.y * 400 + n2) * 2 + 1] = clock64(); ++n2; } } while (0)
#else
#define TM(i)
#define T2(k)
#endif
#ifdef SYN_SPLIT_TIMING  // per-WAVE stamps of workgroup (7, 0): slot (800 + (wave * 48 + n3) * 2 + i) of the int64 buffer
#define T3(i) do { if (g_tm && lane == 0 && blockIdx.x == 7 && blockIdx.y == 0 && n3 < 48) g_tm[1600 + ((tid >> 6) * 48 + n3) * 2 + (i)] = clock64(); } while (0)
#else
#define T3(i)
#endif

// Arithmetic "split9" (synthsr_conv_ctx.arithmetic = 2): all nine partial products a_i b_j instead of six -- an fp32 product is
// then reproduced EXACTLY (tests/test_split_arithmetic_cpu.py) at 1.5x the MFMAs; same packed weights, same kernels (template
// NPROD).  The number of products is an ARGUMENT of the three entry points below (syn_split_fwd / _upfwd / _wgrad, `nprod`); the
// launch helpers of this file read it from t_nprod, which those entry points set on every call -- no state survives a call.
static thread_local int t_nprod = 6;
// Kernel choices that were A/B switches in rounds 3-4 (options 8 and 12 of the former synthsr_conv3d_set_option) and are now
// fixed at what the measurements kept (profiles/r04_split_fwd_variants.txt, r04_split_wgrad_*_ab.txt, r05_split_wgrad_var24_ab.txt):
// forward = conversion inside the K loop (fwd2), the LDS-weights kernel (fwd3) where it quantises better; weight gradient = five
// stacked column tiles for 24 columns, 48-column workgroups where 48 divides Cout with 16 input channels each where 16 divides
// Cin, all 24 input channels in one workgroup where Cin = Cout = 24.

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TZ = 4, TY = 4, TX = 16, HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HVOX = HZ * HY * HX;
constexpr uint32_t OOB = 0x80000000u;
constexpr int PLANE = HVOX * 16;   // one bf16 piece of an 8-channel halo image
constexpr int BUF = 3 * PLANE;     // the three pieces
constexpr int NSTEP27 = 7;         // 27 taps, 4 per MFMA (the 28th slot carries zero weights)
constexpr int SLAB_Z = 4;

// tile schedule: see conv_bf16.hip (tiles enumerated slab by slab, the list cut into 8 contiguous parts, one per XCD)
struct TileWalk {
  int pos, end, stride;
};
// Workgroup (x = b, y, z) of a G-wide launch has linear id b + lin0 (lin0 = G (y + gridDim.y z)) and runs on XCD id % 8; the
// workgroups of one XCD and one (y, z) are those of one class c = b % 8: b = c, c + 8, ... -- G / 8 of them, one more for
// c < G % 8.  The tiles are cut into 8 consecutive segments, one per XCD, each as long as its class has workgroups (so that
// every workgroup gets ntiles / G tiles +- 1 whatever G: with equal segments a G of 85 = 3 classes of 10 and 5 of 11 left 6 % of
// the weight-gradient's time to the small classes, a G of 10 27 %); inside a segment the workgroups of the class interleave, so
// neighbouring tiles are in flight together.  Host-callable: synthsr_split_tile_schedule (tests/test_host_cpu.py walks every
// launch geometry of the network and checks that each tile is visited exactly once).
__host__ __device__ inline TileWalk tile_walk_of(int G, int b, int lin0, int ntiles) {
  TileWalk w;
  if (G >= 8) {
    const int base = G >> 3, rem = G & 7;
    const int k = (b + lin0) & 7;
    int before = 0;  // workgroups (of this y, z) on the XCDs in front of this one
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) before += kk < k ? base + ((((kk - lin0) & 7) < rem) ? 1 : 0) : 0;
    const int nc = base + (((b & 7) < rem) ? 1 : 0);
    w.pos = (int)((int64_t)ntiles * before / G) + (b >> 3);
    w.end = (int)((int64_t)ntiles * (before + nc) / G);
    w.stride = nc;
  } else {  // fewer workgroups than XCDs: plain striding
    w.pos = b;
    w.end = ntiles;
    w.stride = G;
  }
  return w;
}
__device__ __forceinline__ TileWalk tile_walk(int ntiles) {
  return tile_walk_of((int)gridDim.x, (int)blockIdx.x, (int)gridDim.x * (int)(blockIdx.y + gridDim.y * blockIdx.z), ntiles);
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

// bits h = 0 .. n - 1 set where the coordinate o + h lies outside [0, D) -- the out-of-range mask of one axis of a staged box in
// closed form (round 6: the n compare / select / or triplets per axis were ~120 scalar instructions per request)
__device__ __forceinline__ uint32_t syn_oob_bits(int o, int n, int D) {
  const int lo = max(0, -o), hi = min(n - 1, D - 1 - o);
  const uint32_t valid = hi >= lo ? (((2u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
  return ((1u << n) - 1u) & ~valid;
}

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

// the partial products (weight / first-operand piece a, activation / second-operand piece b) in issue order, smallest first:
// six: (2,0) (0,2) (1,1) (1,0) (0,1) (0,0); nine: the three smallest ones (2,2) (2,1) (1,2) in front
constexpr int split_combo_a(int c, int np) {
  if (np == 9) {
    if (c == 0) return 2;
    if (c == 1) return 2;
    if (c == 2) return 1;
    c -= 3;
  }
  return c == 0 ? 2 : ((c == 2 || c == 3) ? 1 : 0);
}
constexpr int split_combo_b(int c, int np) {
  if (np == 9) {
    if (c == 0) return 2;
    if (c == 1) return 1;
    if (c == 2) return 2;
    c -= 3;
  }
  return c == 1 ? 2 : ((c == 2 || c == 4) ? 1 : 0);
}

// ELU(alpha = 1), fp32 accuracy (same function as conv3d.hip: exp2 away from 0, degree-5 Taylor on (-1/8, 0])
__device__ __forceinline__ float elu_f(float v) {
  const float e = __builtin_amdgcn_exp2f(v * 1.44269504088896341f) - 1.f;
  const float p = v * fmaf(v, fmaf(v, fmaf(v, fmaf(v, 1.f / 120.f, 1.f / 24.f), 1.f / 6.f), 0.5f), 1.f);
  const float n = v > -0.125f ? p : e;
  return v > 0.f ? v : n;
}
__device__ __forceinline__ float elu_dy(float y) { return y > 0.f ? 1.f : y + 1.f; }

struct SplitFwdArgs {
  const float* in;
  const u32x4* wp;      // [piece 3][co-chunk][cc][step 7][mt][lane 64] x 8 bf16
  const float* bias;
  const float* addend;  // act 0 / 1: added before the activation (may be `out`); act 2: ELU output of the layer below
  float* out;
  float* stats_partial;  // [gridDim.x][2 Cout] per-workgroup-column sums | sums of squares of the output (ST), or null
  int D0, D1, D2, Cin, Cout, ncc, tiles1, tiles2, ntiles, act;
  int stacked;  // wp holds the stacked 24-channel layout [cc][step 7][tile 5][lane] (conv3d_split_fwd2_kernel<2, ., ., true>)
};

// UPM = 2: data gradient of the up-sampled channel range of a folded decoder conv (unet.py; ext/neuron/models.py:426-444
// UpSampling3D -> concatenate -> Conv3D).  `in` = dz on the 2x grid, D0..D2 = the LOW-resolution grid, output = d(lo).  The 8
// output parities are K chunks (chunk = parity * ncc + input-channel chunk): parity p stages the sub-lattice dz[2 v + p] and
// multiplies by its transposed 8-tap set (2 K steps, taps syn_split_tap8), whose 2x2x2 window starts at halo offset 1 - p per
// axis.  wp = 8 parity sets [piece][co-chunk][cc][step 2][mt][lane], back to back.
template <int MT, bool ST, int UPM = 0, int NPROD = 6>
__global__ __launch_bounds__(256, 2) void conv3d_split_fwd_kernel(const SplitFwdArgs a) {
  static_assert(NPROD == 6 || NPROD == 9, "six partial products, or all nine");
  static_assert(UPM == 0 || UPM == 2, "the folded forward pass has its own kernel");
  static_assert(!(UPM && ST), "statistics belong to plain forward convs");
  constexpr int NSTEP = UPM ? 2 : NSTEP27;
  constexpr int UPS = UPM == 2 ? 2 : 1;  // the staged tensor lives on the 2x grid
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4;
  const int chunk = blockIdx.y, nchunks = gridDim.y;
  const TileWalk walk = tile_walk(a.ntiles);
  const int tiles0 = a.ntiles / (a.tiles1 * a.tiles2);
  const int D0 = a.D0, D1 = a.D1, D2 = a.D2, Cin = a.Cin, Cout = a.Cout, ncc = a.ncc;

  // per-lane LDS byte offset of the tap of K slot 4 s + g (relative to the voxel's place in a piece); the spare slot (zero
  // weights) reads what its pair partner reads.  Lane m holds x-voxel syn_split_voxel(m): see syn_split_tap (bank conflicts)
  int koff[NSTEP];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    if constexpr (UPM) {
      const int t8 = syn_split_tap8(4 * s + g);
      koff[s] = (((t8 >> 2) * HY + ((t8 >> 1) & 1)) * HX + (t8 & 1)) * 16;  // + the parity's window origin, per chunk
    } else {
      int tap = syn_split_tap(4 * s + g);
      if (tap < 0) tap = syn_split_tap((4 * s + g) ^ 1);
      koff[s] = ((tap / 9 * HY + (tap / 3) % 3) * HX + tap % 3) * 16;
    }
  }
  const int xv = syn_split_voxel(m);
  const int lbase = (wave * HY * HX + xv) * 16;  // voxel (z = wave, y = 0, x = xv) of the tile, tap (0, 0, 0)

  // staging: 16-byte piece j = tid + 256 i of the 8-channel halo image -> halo voxel j >> 1, channels 4 (j & 1) .. + 3: the two
  // lanes of a voxel sit in ONE load instruction, which then touches 32 cache lines instead of 64 (the vector memory pipe's
  // tag rate, not bandwidth, is what these 96-byte-strided reads cost)
  // UPM (round 6): a parity's 2x2x2 window at origin o = 1 - p reads the halo coordinates o .. o + 4 (z, y) and o .. o + 16 (x)
  // only -- 5 x 5 x 17 = 425 of the 648 voxels: the staged box is that sub-box (4 instead of 6 pieces per thread: a third less
  // to load, split and store), placed at the origin of the chunk's parity; the rest of the image is never read
  constexpr int SZ = UPM ? HZ - 1 : HZ, SY = UPM ? HY - 1 : HY, SX = UPM ? HX - 1 : HX;
  constexpr int NP = SZ * SY * SX * 2, NL = (NP + 255) / 256;  // plain: 1296 pieces, 6 per thread (the last one only for tid < 16)
  int prel[NL], plds[NL];
  uint32_t pmask[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int j = tid + 256 * i;
    const int v = j >> 1, h = j & 1;
    const int hz = v / (SY * SX), r = v - hz * (SY * SX), hy = r / SX, hx = r - hy * SX;
    prel[i] = ((UPS * hz * (UPS * D1) + UPS * hy) * (UPS * D2) + UPS * hx) * Cin * 4 + h * 16;
    plds[i] = ((hz * HY + hy) * HX + hx) * 16 + h * 8;
    pmask[i] = j < NP ? ((1u << hz) | (1u << (6 + hy)) | (1u << (12 + hx))) : 0xFFFFFFFFu;
  }
  int stg_off = 0;  // UPM: LDS offset of the sub-box of the image in flight (its parity's window origin)
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)((int64_t)(UPM ? 8 : 1) * D0 * D1 * D2 * Cin * 4), 0x00020000);
  f32x4 stg[NL];
  auto load_halo = [&](int t, int cc_) {
    const int par = UPM ? cc_ / ncc : 0, cc = UPM ? cc_ - par * ncc : cc_;
    int z0, y0, x0;
    tile_decode(t, tiles0, a.tiles1, a.tiles2, z0, y0, x0);
    uint32_t bad = 0x80000000u;
    bad |= syn_oob_bits(z0 - 1, HZ, D0) | (syn_oob_bits(y0 - 1, HY, D1) << 6) | (syn_oob_bits(x0 - 1, HX, D2) << 12);
    int base;
    if constexpr (UPM) {
      const int pz = (par >> 2) & 1, py = (par >> 1) & 1, px = par & 1, oz = 1 - pz, oy = 1 - py, ox = 1 - px;
      base = (((2 * (z0 - 1 + oz) + pz) * (2 * D1) + (2 * (y0 - 1 + oy) + py)) * (2 * D2) + (2 * (x0 - 1 + ox) + px)) * Cin * 4 +
             cc * 32;
      stg_off = ((oz * HY + oy) * HX + ox) * 16;
      // the sub-box's coordinate j is halo coordinate j + o: the out-of-range bits move down by the origin, field by field
      bad = 0x80000000u | ((bad & 0x3Fu) >> oz) | ((((bad >> 6) & 0x3Fu) >> oy) << 6) | ((((bad >> 12) & 0x3FFFFu) >> ox) << 12);
    } else {
      base = ((((z0 - 1) * D1 + (y0 - 1)) * D2 + (x0 - 1)) * Cin + cc * 8) * 4;
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const uint32_t vo = (pmask[i] & bad) ? OOB : (uint32_t)(prel[i] + base);
      stg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (int)vo, 0, 0));
    }
  };
  auto store_halo = [&](int buf) {  // four fp32 -> 3 x (four bf16 = 8 bytes)
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    unsigned char* dst = lds + buf * BUF + stg_off;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      if (i == NL - 1 && tid + 256 * i >= NP) continue;
      uint32_t p0, p1, p2, q0, q1, q2;
      syn_split3(stg[i][0], stg[i][1], p0, p1, p2);
      syn_split3(stg[i][2], stg[i][3], q0, q1, q2);
      *reinterpret_cast<u32x2*>(dst + plds[i]) = (u32x2){p0, q0};
      *reinterpret_cast<u32x2*>(dst + PLANE + plds[i]) = (u32x2){p1, q1};
      *reinterpret_cast<u32x2*>(dst + 2 * PLANE + plds[i]) = (u32x2){p2, q2};
    }
  };

  // weight fragments of this co-chunk: piece q at wq[q]; fragment (cc, step, mt) at ((cc * NSTEP + step) * MT + mt) * 64
  const int64_t piece_stride = (int64_t)nchunks * ncc * NSTEP * MT * 64;
  const int64_t par_stride = 3 * piece_stride;  // UPM: one parity's weight set
  const u32x4* __restrict__ wbase = a.wp + (int64_t)chunk * ncc * NSTEP * MT * 64 + lane;
  const int nck = UPM ? 8 * ncc : ncc;          // K chunks per tile

  float s1[ST ? MT : 1][4], s2[ST ? MT : 1][4];  // BatchNorm partial sums of this lane's channels (ST only)
#pragma unroll
  for (int mt = 0; mt < (ST ? MT : 1); ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) s1[mt][i] = s2[mt][i] = 0.f;

  const int64_t out_bytes = (int64_t)D0 * D1 * D2 * Cout * 4;
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t radd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.addend ? a.addend : a.out), 0, (int)out_bytes, 0x00020000);

  // Register budget (2 workgroups per CU: 256 VGPRs): ONE set of weight fragments and ONE set of activation fragments.  The six
  // products of a K step are ordered so that every piece's registers can be re-loaded for the next step right after their last
  // use and are not needed again for as long as possible:
  //     (w0,x2) (w0,x1) (w0,x0) | reload w0   (w1,x0) (w1,x1) | reload w1, x1   (w2,x0) | reload w2, x0      [x2 after its product]
  // weights (L2 latency) get >= 3 products = 12 MT MFMAs of cover, activations (LDS) >= 2; the weight stream runs on across
  // chunk and tile boundaries (the next chunk's first fragments are requested during the last step of this one).
  u32x4 wa[3][MT], xb[3][TY];
  auto wload = [&](const u32x4* wf, int s, int q) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) wa[q][mt] = wf[q * piece_stride + (s * MT + mt) * 64];
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;

  int buf = 0;
  if (walk.pos < walk.end) {
    load_halo(walk.pos, 0);
    store_halo(0);
    wload(wbase, 0, 0);
    wload(wbase, 0, 1);
    wload(wbase, 0, 2);
  }
  int tix = -1;
  (void)tix;
  for (int t = walk.pos; t < walk.end; t += walk.stride) {
    f32x4 acc[TY][MT];
#pragma unroll
    for (int y = 0; y < TY; ++y)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[y][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int cc = 0; cc < nck; ++cc) {
      ++tix;
      TM(0);
      __syncthreads();  // image `buf` is complete; nobody reads the other one any more
      TM(1);
      const bool more = cc + 1 < nck || t + walk.stride < walk.end;
      if (cc + 1 < nck) load_halo(t, cc + 1);
      else if (t + walk.stride < walk.end) load_halo(t + walk.stride, 0);
      TM(2);
      // fragments of this K chunk and step 0 of the one that follows (UPM: chunk = parity * ncc + c)
      auto wchunk = [&](int c_) {
        const int par = UPM ? c_ / ncc : 0;
        return wbase + (int64_t)par * par_stride + (int64_t)(c_ - par * ncc) * NSTEP * MT * 64;
      };
      const u32x4* wf = wchunk(cc);
      const u32x4* wf_next = wchunk(cc + 1 < nck ? cc + 1 : 0);
      int win = 0;  // UPM: the parity's 2x2x2 window starts at halo offset 1 - p per axis
      if constexpr (UPM) {
        const int par = cc / ncc;
        win = (((1 - ((par >> 2) & 1)) * HY + (1 - ((par >> 1) & 1))) * HX + (1 - (par & 1))) * 16;
      }
      const unsigned char* img = lds + buf * BUF + lbase + win;
      auto xload = [&](int s, int q) {  // piece q of the four x-rows at the taps of step s
#pragma unroll
        for (int y = 0; y < TY; ++y) xb[q][y] = *reinterpret_cast<const u32x4*>(img + q * PLANE + koff[s] + y * (HX * 16));
      };
      auto mma = [&](auto QA, auto QB) {  // acc += (weight piece QA) x (activation piece QB)
        constexpr int qa = decltype(QA)::value, qb = decltype(QB)::value;
#pragma unroll
        for (int y = 0; y < TY; ++y)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[y][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[qa][mt]),
                                                                 __builtin_bit_cast(bf16x8, xb[qb][y]), acc[y][mt], 0, 0, 0);
      };
      xload(0, 2);
      xload(0, 1);
      xload(0, 0);
      sfor<0, NSTEP>([&](auto S) {
        constexpr int s = decltype(S)::value;
        constexpr bool last = s + 1 == NSTEP;
        const u32x4* wn = last ? wf_next : wf;
        constexpr int sn = last ? 0 : s + 1;
        if constexpr (NPROD == 9) {  // all nine products: a weight piece is re-loaded after its three products, activations at the end
          __builtin_amdgcn_sched_barrier(0);
          mma(I0{}, I2{});
          mma(I0{}, I1{});
          mma(I0{}, I0{});
          __builtin_amdgcn_sched_barrier(0);
          wload(wn, sn, 0);
          __builtin_amdgcn_sched_barrier(0);
          mma(I1{}, I2{});
          mma(I1{}, I1{});
          mma(I1{}, I0{});
          __builtin_amdgcn_sched_barrier(0);
          wload(wn, sn, 1);
          __builtin_amdgcn_sched_barrier(0);
          mma(I2{}, I2{});
          mma(I2{}, I1{});
          mma(I2{}, I0{});
          __builtin_amdgcn_sched_barrier(0);
          wload(wn, sn, 2);
          if constexpr (!last) {
            xload(sn, 2);
            xload(sn, 1);
            xload(sn, 0);
          }
          return;
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(I0{}, I2{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!last) xload(sn, 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(I0{}, I1{});
        __builtin_amdgcn_sched_barrier(0);
        mma(I0{}, I0{});
        __builtin_amdgcn_sched_barrier(0);
        wload(wn, sn, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(I1{}, I0{});
        __builtin_amdgcn_sched_barrier(0);
        mma(I1{}, I1{});
        __builtin_amdgcn_sched_barrier(0);
        wload(wn, sn, 1);
        if constexpr (!last) xload(sn, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(I2{}, I0{});
        __builtin_amdgcn_sched_barrier(0);
        wload(wn, sn, 2);
        if constexpr (!last) xload(sn, 0);
      });
      __builtin_amdgcn_sched_barrier(0);
      TM(3);
#ifdef SYN_SPLIT_TIMING
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      TM(6);
#endif
      if (more) store_halo(buf ^ 1);
      TM(4);
      buf ^= 1;
    }
    // ---- epilogue: lane (m, g): channels (chunk*MT + mt)*16 + 4g + i of voxel (z0 + wave, y0 + y, x0 + xv)
    int z0, y0, x0;
    tile_decode(t, tiles0, a.tiles1, a.tiles2, z0, y0, x0);
    const int gz = z0 + wave, gx = x0 + xv;
    const bool zx_ok = gz < D0 && gx < D2;
    const uint32_t row0 = (uint32_t)((gz * D1 + y0) * D2 + gx) * (uint32_t)(Cout * 4);
    const uint32_t ystep = (uint32_t)(D2 * Cout * 4);
    auto epi = [&](auto ACTC, auto ADDC) {
      constexpr int ACT = decltype(ACTC)::value;
      constexpr bool ADD = decltype(ADDC)::value;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int co = (chunk * MT + mt) * 16 + 4 * g;
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (a.bias && co < Cout) bias = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
        for (int y = 0; y < TY; ++y) {
          const bool vok = zx_ok && (y0 + y) < D1;
          const uint32_t off = (vok && co < Cout) ? row0 + y * ystep + (uint32_t)(co * 4) : OOB;
          f32x4 v;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc[y][mt][i] + bias[i];
          if constexpr (ADD || ACT == 2) {
            const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(radd, (int)off, 0, 0));
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ACT == 2 ? v[i] * elu_dy(b[i]) : v[i] + b[i];
          }
          if constexpr (ACT == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = elu_f(v[i]);
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rout, (int)off, 0, 0);
          if constexpr (ST) {
            const float w = off != OOB ? 1.f : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float r = w * v[i];
              s1[mt][i] += r;
              s2[mt][i] += r * r;
            }
          }
        }
      }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if constexpr (ST) {  // forward layers in front of a BatchNorm: no addend
      if (a.act == 1) epi(std::integral_constant<int, 1>{}, F_{});
      else epi(std::integral_constant<int, 0>{}, F_{});
    } else {
      if (a.act == 2) epi(std::integral_constant<int, 2>{}, F_{});
      else if (a.addend) {
        if (a.act == 1) epi(std::integral_constant<int, 1>{}, T_{});
        else epi(std::integral_constant<int, 0>{}, T_{});
      } else if (a.act == 1) epi(std::integral_constant<int, 1>{}, F_{});
      else epi(std::integral_constant<int, 0>{}, F_{});
    }
    TM(5);
  }
  if constexpr (ST) {
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
    // partial[workgroup column][2 Cout] (sums | sums of squares), the layout of synthsr_bn_stats_from_partials
    float* dst = a.stats_partial + (int64_t)blockIdx.x * (2 * Cout);
    for (int e = tid; e < MT * 16; e += 256) {
      const int c = chunk * MT * 16 + e;
      if (c < Cout) {
        dst[c] = red[e] + red[2 * MT * 16 + e] + red[4 * MT * 16 + e] + red[6 * MT * 16 + e];
        dst[Cout + c] = red[MT * 16 + e] + red[3 * MT * 16 + e] + red[5 * MT * 16 + e] + red[7 * MT * 16 + e];
      }
    }
  }
}

// ---- round 4: the forward / data-gradient kernel with the conversion of the next halo image INSIDE the K loop.
// s_memtime stamps of the kernel above (profiles/r04_split_fwd_phase_cycles_before.txt): per 8-channel chunk a wave spends
// 8 400 cycles in its K loop (336 MFMAs = 5 376 cycles of matrix pipe), 3 000-4 000 converting the next halo image and, once per
// tile, 7 000-10 000 in the epilogue.  Here piece i of a thread's share of the next image is converted in K step i + 1, between
// the MFMA groups (its loads travel in two register groups: 16 instead of 24 staging registers), the bias sits in LDS, and the
// LDS planes are padded to 6 x 256 staging pieces so that every lane stores (no divergent tail): 217 -> 176 registers at MT = 2
// and 5-15 % less time on the layers with many tiles.  (Parking a tile's accumulators and running its epilogue inside the next
// tile's K loop was built too: two copies of the K loop behind an if / else cost ~100 spilled registers, and without that the
// gain was nil -- profiles/r04_split_fwd2_ablation.txt: what is left is exposed memory latency and the instruction mix, not
// phases.)  Activation / addend handling is a template parameter: EPI 0 linear, 1 ELU, 2 x ELU'(addend), 3 + addend,
// 4 ELU(. + addend).
//
// STK (round 4, the Cout = 24 layers: 160^3, the matrix pipe's largest customers): the three bf16 pieces of the weights are stacked
// along M instead of padding 24 output channels to two 16-row tiles per piece.  The six products a0 b0 + a0 b1 + a1 b0 + a1 b1 +
// a0 b2 + a2 b0 need, per activation piece, the weight pieces {a0, a1, a2} (b0), {a0, a1} (b1), {a0} (b2) = 72 + 48 + 24 rows:
// with the row tiles  T0 = a0[0..15]  T1 = a1[0..15]  T2 = a2[0..15]  T3 = a0[16..23] | a1[16..23]  T4 = a2[16..23] | 0
// that is b0 x {T0..T4}, b1 x {T0, T1, T3}, b2 x {T0, T3} = 10 MFMAs per K step and voxel row instead of 12 (T3 x b2 also forms
// a1 b2 for channels 16..23: one of the three omitted products, exact like the others).  Each tile has its own accumulator;
// the epilogue adds T0 + T1 + T2 (same lanes) and, for channels 16..23, the two halves of T3 (rows 8..15 sit 32 lanes up) + T4.
// Every accumulator still receives its products smallest first (b2, b1, b0).
constexpr int PLANE2 = 6 * 256 * 8;  // bytes of one bf16 piece of an 8-channel halo image, padded (>= HVOX * 16)
constexpr int BUF2 = 3 * PLANE2;
static_assert(PLANE2 >= PLANE, "padded plane holds the halo image");
constexpr int STK_TILES = 5;

// KS = 2 (round 6; launches with at most one workgroup per CU, i.e. the 20^3 layers: 200 or 100 units on 256 CUs): a 512-thread
// workgroup whose two halves (waves 0-3 | 4-7) take the EVEN | ODD input-channel chunks of the same tile and co-chunk, each half
// with its own double-buffered image; at the end of a tile the halves exchange half of their accumulators through LDS (the image
// buffer that was consumed last) and each runs the epilogue of two of the four y rows.  The serial chain of a workgroup halves
// (24 -> 12 chunks at 192 input channels) without staging anything twice, and every SIMD holds two waves instead of one.
template <int MT, bool ST, int EPI, bool STK, int KS = 1>
__global__ __launch_bounds__(256 * KS, KS == 1 ? 2 : 1) void conv3d_split_fwd2_kernel(const SplitFwdArgs a) {
  static_assert(KS == 1 || (KS == 2 && !STK), "split-K halves: the plain layout");
  constexpr int NSTEP = NSTEP27;
  constexpr int TYE = TY / KS;              // y rows whose epilogue a wave runs (KS = 2: each half takes two of the four)
  constexpr int NITEM = TYE * MT;           // epilogue items: one f32x4 (4 channels of a voxel) per lane each
  constexpr int NWT = STK ? STK_TILES : MT; // weight row tiles per piece set (STK: of the stacked set) = accumulators per row
  static_assert(!ST || EPI <= 1, "statistics belong to plain forward convs");
  static_assert(!STK || MT == 2, "the stacked layout is the 24-channel one (two output tiles)");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int half = KS == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);   // wave-uniform
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;   // roles inside the half
  unsigned char* const ldsh = lds + half * (2 * BUF2);                   // the half's own pair of images
  const int m = lane & 15, g = lane >> 4;
  const int chunk = blockIdx.y, nchunks = gridDim.y;
  const TileWalk walk = tile_walk(a.ntiles);
  const int tiles0 = a.ntiles / (a.tiles1 * a.tiles2);
  const int D0 = a.D0, D1 = a.D1, D2 = a.D2, Cin = a.Cin, Cout = a.Cout, ncc = a.ncc;

  int koff[NSTEP];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    int tap = syn_split_tap(4 * s + g);
    if (tap < 0) tap = syn_split_tap((4 * s + g) ^ 1);
    koff[s] = ((tap / 9 * HY + (tap / 3) % 3) * HX + tap % 3) * 16;
  }
  const int xv = syn_split_voxel(m);
  const int lbase = (wave * HY * HX + xv) * 16;

  constexpr int NP = HVOX * 2, NL = 6;
  static_assert(NL * 256 >= NP, "six staging pieces per thread cover the halo image");
  int prel[NL];
  uint32_t pmask[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int j = tid + 256 * i;
    const int v = j >> 1, h = j & 1;
    const int hz = v / (HY * HX), r = v - hz * (HY * HX), hy = r / HX, hx = r - hy * HX;
    prel[i] = ((hz * D1 + hy) * D2 + hx) * Cin * 4 + h * 16;
    pmask[i] = j < NP ? ((1u << hz) | (1u << (6 + hy)) | (1u << (12 + hx))) : 0xFFFFFFFFu;
  }
  const int plds0 = (tid >> 1) * 16 + (tid & 1) * 8;  // piece i sits 128 voxels = 2048 bytes further on
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)((int64_t)D0 * D1 * D2 * Cin * 4), 0x00020000);
  // the six pieces of a thread travel in two groups (registers): pieces 0-3 are requested at the top of a chunk and converted
  // in K steps 1-4, pieces 4-5 are requested in step 2 (into the registers pieces 0-1 have left) and converted in steps 5-6
  f32x4 stg[NL];
  uint32_t hbad = 0;  // out-of-range mask of the halo image in flight (wave-uniform)
  int hbase = 0;      // its byte offset
  // the tile part of a request (origin, out-of-range mask: two divisions and thirty compare / select / or triplets) is computed when
  // the requested tile CHANGES, not once per chunk (round 6: ~170 scalar instructions per call, 3 ... 12 calls per tile)
  int hw_tile = -1, hw_base0 = 0;
  uint32_t hw_bad = 0;
  auto halo_where = [&](int t, int cc, bool none) {  // none: there is no next chunk -- every lane reads out of range (zeros)
    if (t != hw_tile) {
      int z0, y0, x0;
      tile_decode(t, tiles0, a.tiles1, a.tiles2, z0, y0, x0);
      uint32_t bad = 0x80000000u;
      bad |= syn_oob_bits(z0 - 1, HZ, D0) | (syn_oob_bits(y0 - 1, HY, D1) << 6) | (syn_oob_bits(x0 - 1, HX, D2) << 12);
      hw_bad = bad;
      hw_base0 = (((z0 - 1) * D1 + (y0 - 1)) * D2 + (x0 - 1)) * Cin * 4;
      hw_tile = t;
    }
    hbad = none ? 0xFFFFFFFFu : hw_bad;
    hbase = hw_base0 + cc * 32;
  };
  auto load_pieces = [&](auto I0_, auto I1_) {
    constexpr int i0 = decltype(I0_)::value, i1 = decltype(I1_)::value;
#pragma unroll
    for (int i = i0; i < i1; ++i) {
      const uint32_t vo = (pmask[i] & hbad) ? OOB : (uint32_t)(prel[i] + hbase);
#if SYN_ABL & 1
      stg[i] = (f32x4){__uint_as_float(vo), 1.f, 2.f, 3.f};
#else
      stg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (int)vo, 0, 0));
#endif
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P4 = std::integral_constant<int, 4>;
  using P6 = std::integral_constant<int, 6>;
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  uint32_t cp[6];  // the piece being converted: three bf16 pairs of its first / second two values
#if SYN_ABL & 2
  auto conv_a = [&](int i) { cp[0] = __float_as_uint(stg[i][0]); cp[1] = __float_as_uint(stg[i][1]); cp[2] = cp[0] ^ cp[1]; };
  auto conv_b = [&](int i) { cp[3] = __float_as_uint(stg[i][2]); cp[4] = __float_as_uint(stg[i][3]); cp[5] = cp[3] ^ cp[4]; };
#else
  auto conv_a = [&](int i) { syn_split3(stg[i][0], stg[i][1], cp[0], cp[1], cp[2]); };
  auto conv_b = [&](int i) { syn_split3(stg[i][2], stg[i][3], cp[3], cp[4], cp[5]); };
#endif
  auto conv_c = [&](int i, unsigned char* dst) {
#if SYN_ABL & 4
    if (cp[0] == 0x12345u && cp[5] == 0x54321u)  // (never)
#endif
    {
    *reinterpret_cast<u32x2*>(dst + plds0 + i * 2048) = (u32x2){cp[0], cp[3]};
    *reinterpret_cast<u32x2*>(dst + PLANE2 + plds0 + i * 2048) = (u32x2){cp[1], cp[4]};
    *reinterpret_cast<u32x2*>(dst + 2 * PLANE2 + plds0 + i * 2048) = (u32x2){cp[2], cp[5]};
    }
  };

  // weight fragments.  Plain layout: piece q at q * piece_stride, fragment (cc, step, mt) at ((cc * NSTEP + step) * MT + mt) * 64;
  // stacked layout (STK): one set [cc][step][tile 5][lane]
  const int64_t piece_stride = (int64_t)nchunks * ncc * NSTEP * MT * 64;
  const u32x4* __restrict__ wbase = a.wp + (STK ? 0 : (int64_t)chunk * ncc * NSTEP * MT * 64) + lane;
  constexpr int WCHUNK = NSTEP * NWT * 64;  // fragments (u32x4) of one input-channel chunk in one piece set

  float s1[ST ? MT : 1][4], s2[ST ? MT : 1][4];
#pragma unroll
  for (int mt = 0; mt < (ST ? MT : 1); ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) s1[mt][i] = s2[mt][i] = 0.f;

  const int64_t out_bytes = (int64_t)D0 * D1 * D2 * Cout * 4;
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t radd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.addend ? a.addend : a.out), 0, (int)out_bytes, 0x00020000);
  const uint32_t ystep = (uint32_t)(D2 * Cout * 4);

  // the bias of this co-chunk sits in LDS behind the images (an epilogue item reads its four channels from there)
  float* lbias = reinterpret_cast<float*>(lds + KS * 2 * BUF2);
  if ((int)threadIdx.x < MT * 16) {
    const int co = chunk * MT * 16 + tid;
    lbias[tid] = (a.bias && co < Cout) ? a.bias[co] : 0.f;
  }

  // Register budget (2 workgroups per CU: 256 VGPRs): ONE set of weight fragments and ONE set of activation fragments; every
  // fragment is re-loaded for the next step right after its last use and then not needed for >= 12 (plain) / 20 (STK) MFMAs
  u32x4 wa[STK ? 1 : 3][STK ? STK_TILES : MT], xb[3][TY];
  bool wfirst = true;
  (void)wfirst;
  auto wload = [&](const u32x4* wf, int s, int q) {  // plain: piece q of step s
#if SYN_ABL & 8
    if (!wfirst) return;
#endif
    if constexpr (!STK) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) wa[q][mt] = wf[q * piece_stride + (s * MT + mt) * 64];
    }
  };
  auto wload_t = [&](const u32x4* wf, int s, int tl) {  // STK: row tile tl of step s
#if SYN_ABL & 8
    if (!wfirst) return;
#endif
    if constexpr (STK) wa[0][tl] = wf[(s * STK_TILES + tl) * 64];
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;

  f32x4 acc[TY][NWT];
  // addend requests in flight ahead of the item being worked on: three (round 6: one left the HBM round trip of most of the eight
  // items exposed; 160^3 data gradient 0.767 -> 0.751 ms, bit-identical, profiles/r06_epilogue_addend_prefetch_ab.txt; seven: no more)
  constexpr int EA = EPI >= 2 ? 3 : 1, ENB = EA + 1;
  f32x4 ev, eb[ENB];
  uint32_t eoff[ENB];
  // epilogue item j = (mt, y) of the finished tile, in three segments; the addends of items j + 1 .. j + EA are requested before
  // item j is worked on (their HBM latency hides behind the items' ELU / stores instead of being exposed eight times per tile)
  auto epi0 = [&](int j, uint32_t row0, uint32_t yok) {
    const int mt = j / TYE, y = j % TYE;
    const int co = (chunk * MT + mt) * 16 + 4 * g;
#if SYN_ABL & 32
    yok = 0;
#endif
    eoff[j % ENB] = (((yok >> y) & 1u) && co < Cout) ? row0 + (uint32_t)y * ystep + (uint32_t)(co * 4) : OOB;   // (row0, yok: of the half's rows)
    if constexpr (EPI >= 2)
      eb[j % ENB] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(radd, (int)eoff[j % ENB], 0, 0));
  };
  auto epi1 = [&](int j) {
    const int mt = j / TYE, y = j % TYE;
    const f32x4 bj = *reinterpret_cast<const f32x4*>(lbias + mt * 16 + 4 * g);
    f32x4 v;
    if constexpr (STK) {
      if (mt == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (acc[y][0][i] + acc[y][1][i]) + acc[y][2][i];
      } else {  // channels 16 + 4 g + i (g < 2): the a0 part here, the a1 part in the lane 32 up (rows 8..15 of T3), a2 in T4
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (acc[y][3][i] + __shfl_down(acc[y][3][i], 32, 64)) + acc[y][4][i];
      }
    } else {
      v = acc[y][mt];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += bj[i];
    if constexpr (EPI == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] *= elu_dy(eb[j % ENB][i]);
    }
    if constexpr (EPI == 3 || EPI == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += eb[j % ENB][i];
    }
    if constexpr (EPI == 1 || EPI == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = elu_f(v[i]);
    }
    ev = v;
  };
  auto epi2 = [&](int j) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ev), rout, (int)eoff[j % ENB], 0, 0);
    if constexpr (ST) {
      const int mt = j / TYE;
      const float w = eoff[j % ENB] != OOB ? 1.f : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float r = w * ev[i];
        s1[mt][i] += r;
        s2[mt][i] += r * r;
      }
    }
  };

  int buf = 0;
  int n2 = 0;
  (void)n2;
  T2(0);
  if (walk.pos < walk.end) {
    halo_where(walk.pos, half, false);
    load_pieces(P0{}, P6{});
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      conv_a(i);
      conv_b(i);
      conv_c(i, ldsh);
    }
    if constexpr (STK) {
#pragma unroll
      for (int tl = 0; tl < STK_TILES; ++tl) wload_t(wbase, 0, tl);
    } else {
      wload(wbase + (int64_t)half * WCHUNK, 0, 0);
      wload(wbase + (int64_t)half * WCHUNK, 0, 1);
      wload(wbase + (int64_t)half * WCHUNK, 0, 2);
    }
    wfirst = false;
  }
  T2(1);
  for (int t = walk.pos; t < walk.end; t += walk.stride) {
#pragma unroll
    for (int y = 0; y < TY; ++y)
#pragma unroll
      for (int tl = 0; tl < NWT; ++tl) acc[y][tl] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int cc = half; cc < ncc; cc += KS) {   // (KS = 2: ncc is even, both halves run the same number of chunks)
      T2(2);
      __syncthreads();  // image `buf` is complete; nobody reads the other one any more
      T2(3);
      const bool last_cc = cc + KS >= ncc;
      const bool more = !last_cc || t + walk.stride < walk.end;
      halo_where(last_cc ? (more ? t + walk.stride : t) : t, last_cc ? half : cc + KS, !more);
      load_pieces(P0{}, P4{});
      const u32x4* wf = wbase + (int64_t)cc * WCHUNK;
      const u32x4* wf_next = wbase + (int64_t)(last_cc ? half : cc + KS) * WCHUNK;
      const unsigned char* img = ldsh + buf * BUF2 + lbase;
      unsigned char* nimg = ldsh + (buf ^ 1) * BUF2;
      auto xload = [&](int s, int q) {
#if SYN_ABL & 16
        if (s != 0) return;
#endif
#pragma unroll
        for (int y = 0; y < TY; ++y) xb[q][y] = *reinterpret_cast<const u32x4*>(img + q * PLANE2 + koff[s] + y * (HX * 16));
      };
      auto mma = [&](auto QA, auto QB) {  // plain: acc += (weight piece QA) x (activation piece QB)
        constexpr int qa = decltype(QA)::value, qb = decltype(QB)::value;
        if constexpr (!STK) {
#pragma unroll
          for (int y = 0; y < TY; ++y)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#if SYN_ABL & 64
              acc[y][mt][0] += __uint_as_float(wa[qa][mt][0] ^ xb[qb][y][0]);
#else
              acc[y][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[qa][mt]),
                                                                   __builtin_bit_cast(bf16x8, xb[qb][y]), acc[y][mt], 0, 0, 0);
#endif
        }
      };
      auto mmt = [&](auto TL, auto QB) {  // STK: acc[.][tile TL] += (row tile TL) x (activation piece QB)
        constexpr int tl = decltype(TL)::value, qb = decltype(QB)::value;
        if constexpr (STK) {
#pragma unroll
          for (int y = 0; y < TY; ++y)
#if SYN_ABL & 64
            acc[y][tl][0] += __uint_as_float(wa[0][tl][0] ^ xb[qb][y][0]);
#else
            acc[y][tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[0][tl]),
                                                                 __builtin_bit_cast(bf16x8, xb[qb][y]), acc[y][tl], 0, 0, 0);
#endif
        }
      };
      xload(0, 2);
      xload(0, 1);
      xload(0, 0);
      sfor<0, NSTEP>([&](auto S) {
        constexpr int s = decltype(S)::value;
        constexpr bool last = s + 1 == NSTEP;
        const u32x4* wn = last ? wf_next : wf;
        constexpr int sn = last ? 0 : s + 1;
        constexpr int ci = s - 1;  // staging piece converted in this step
        constexpr bool cv = ci >= 0 && ci < NL;
        if constexpr (STK) {
          // T0 and T3 (needed first in the next step) finish first: T0 b2, T3 b2, T0 b1, T3 b1, T0 b0, T3 b0 | T1 b1, T1 b0 |
          // T2 b0, T4 b0 -- every fragment is re-loaded >= 20 MFMAs before its next use, every accumulator gets b2, b1, b0
          __builtin_amdgcn_sched_barrier(0);
          mmt(I0{}, I2{});
          mmt(I3{}, I2{});
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (!last) xload(sn, 2);
          if constexpr (cv) conv_a(ci);
          __builtin_amdgcn_sched_barrier(0);
          mmt(I0{}, I1{});
          mmt(I3{}, I1{});
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (cv) conv_b(ci);
          __builtin_amdgcn_sched_barrier(0);
          mmt(I0{}, I0{});
          __builtin_amdgcn_sched_barrier(0);
          wload_t(wn, sn, 0);
          if constexpr (cv) conv_c(ci, nimg);
          __builtin_amdgcn_sched_barrier(0);
          mmt(I3{}, I0{});
          __builtin_amdgcn_sched_barrier(0);
          wload_t(wn, sn, 3);
          if constexpr (s == 2) load_pieces(P4{}, P6{});
          __builtin_amdgcn_sched_barrier(0);
          mmt(I1{}, I1{});
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (!last) xload(sn, 1);
          __builtin_amdgcn_sched_barrier(0);
          mmt(I1{}, I0{});
          __builtin_amdgcn_sched_barrier(0);
          wload_t(wn, sn, 1);
          __builtin_amdgcn_sched_barrier(0);
          mmt(I2{}, I0{});
          mmt(I4{}, I0{});
          __builtin_amdgcn_sched_barrier(0);
          wload_t(wn, sn, 2);
          wload_t(wn, sn, 4);
          if constexpr (!last) xload(sn, 0);
        } else {
          __builtin_amdgcn_sched_barrier(0);
          mma(I0{}, I2{});
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (!last) xload(sn, 2);
          if constexpr (cv) conv_a(ci);
          __builtin_amdgcn_sched_barrier(0);
          mma(I0{}, I1{});
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (cv) conv_b(ci);
          __builtin_amdgcn_sched_barrier(0);
          mma(I0{}, I0{});
          __builtin_amdgcn_sched_barrier(0);
          wload(wn, sn, 0);
          if constexpr (cv) conv_c(ci, nimg);
          __builtin_amdgcn_sched_barrier(0);
          mma(I1{}, I0{});
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (s == 2) load_pieces(P4{}, P6{});
          __builtin_amdgcn_sched_barrier(0);
          mma(I1{}, I1{});
          __builtin_amdgcn_sched_barrier(0);
          wload(wn, sn, 1);
          if constexpr (!last) xload(sn, 1);
          __builtin_amdgcn_sched_barrier(0);
          mma(I2{}, I0{});
          __builtin_amdgcn_sched_barrier(0);
          wload(wn, sn, 2);
          if constexpr (!last) xload(sn, 0);
        }
      });
      __builtin_amdgcn_sched_barrier(0);
      T2(4);
      buf ^= 1;
    }
    // ---- epilogue: lane (m, g): channels (chunk*MT + mt)*16 + 4g + i of voxel (z0 + wave, y0 + y, x0 + xv)
    int z0, y0, x0;
    tile_decode(t, tiles0, a.tiles1, a.tiles2, z0, y0, x0);
    if constexpr (KS == 2) {
      // the halves' partial sums meet: rows 2, 3 of half 0 and rows 0, 1 of half 1 travel through the image buffer that was
      // consumed last (`buf ^ 1` after the flip: the NEXT tile's second chunk will be converted into it after the next barrier;
      // `buf` already holds the next tile's first chunk), and half 1 ends with its rows 2, 3 in the slots the epilogue reads
      __syncthreads();  // every wave has left the K loop: nobody reads that image any more
      f32x4* xch = reinterpret_cast<f32x4*>(ldsh + (buf ^ 1) * BUF2) + tid;
#pragma unroll
      for (int y = 0; y < TYE; ++y)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xch[(y * MT + mt) * 256] = half ? acc[y][mt] : acc[y + TYE][mt];
      __syncthreads();
      const f32x4* rcv = reinterpret_cast<const f32x4*>(lds + (half ^ 1) * (2 * BUF2) + (buf ^ 1) * BUF2) + tid;
#pragma unroll
      for (int y = 0; y < TYE; ++y)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x4 mine = half ? acc[y + TYE][mt] : acc[y][mt];
          const f32x4 theirs = rcv[(y * MT + mt) * 256];
          acc[y][mt] = half ? theirs + mine : mine + theirs;   // even chunks + odd chunks, the same order in both halves
        }
    }
    const int gz = z0 + wave, gx = x0 + xv;
    const bool zx_ok = gz < D0 && gx < D2;
    const int yh = y0 + half * TYE;   // first y row of this half's epilogue
    const uint32_t row0 = (uint32_t)((gz * D1 + yh) * D2 + gx) * (uint32_t)(Cout * 4);
    uint32_t yok = 0;
#pragma unroll
    for (int y = 0; y < TYE; ++y) yok |= (zx_ok && (yh + y) < D1) ? (1u << y) : 0u;
    T2(5);
#pragma unroll
    for (int j = 0; j < EA && j < NITEM; ++j) epi0(j, row0, yok);
#pragma unroll
    for (int j = 0; j < NITEM; ++j) {
      if (j + EA < NITEM) epi0(j + EA, row0, yok);
      epi1(j);
      epi2(j);
    }
    T2(6);
  }
  if constexpr (ST) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);  // [wave of the workgroup][2][MT*16]
    const int wv = (int)threadIdx.x >> 6;
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
          red[(wv * 2 + 0) * (MT * 16) + mt * 16 + 4 * g + i] = x1;
          red[(wv * 2 + 1) * (MT * 16) + mt * 16 + 4 * g + i] = x2;
        }
      }
    __syncthreads();
    float* dst = a.stats_partial + (int64_t)blockIdx.x * (2 * Cout);
    for (int e = threadIdx.x; e < MT * 16; e += 256 * KS) {
      const int c = chunk * MT * 16 + e;
      if (c < Cout) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4 * KS; ++w) {   // (KS = 1: ((w0 + w1) + w2) + w3, the order of rounds 4-5)
          t1 += red[(2 * w) * (MT * 16) + e];
          t2 += red[(2 * w + 1) * (MT * 16) + e];
        }
        dst[c] = t1;
        dst[Cout + c] = t2;
      }
    }
  }
}

// ---- round 4, second step: no operand of an MFMA waits on the vector-memory counter.
// Ablation builds of the kernel above (profiles/r04_split_fwd2_ablation.txt, 160^3 24->24): without its MFMAs it still takes
// 0.66 ms, its MFMAs alone 0.59 ms, both together 0.96 ms.  vmcnt counts in order: a weight fragment (L2, requested one K step
// ahead) cannot be waited for without also waiting for every older request of the wave -- the halo loads of the next chunk (HBM)
// and the previous epilogue's stores -- so every chunk exposed one or two HBM round trips, and interleaving instructions could
// not hide them.  Here the matrix operands come from LDS only (lgkmcnt): one 512-thread workgroup per CU; per 8-channel chunk the
// weight fragments of the co-chunk (3 pieces x 7 steps x MT KB) are copied L2 -> registers -> LDS one chunk ahead (both buffers
// when they fit, MT <= 2; one buffer and a second barrier otherwise), the halo image is requested a chunk ahead, converted in the
// last K steps of the chunk before its own and written to the other image buffer; the previous tile's epilogue rides in the
// first chunk of the next tile.  The only waits on vmcnt are for requests that are at least four K steps old, and stores are
// never waited for.  wave = (z plane, y half) of the 4x4x16 tile: two x-rows, fragments of step s + 1 are read (second register
// set) while step s multiplies.  Same products in the same order per accumulator as the kernels above: bit-identical results.
template <int MT>
struct F3Cfg {
  static constexpr int NWP = 3 * NSTEP27 * MT * 64;    // 16-byte pieces of one chunk's weight fragments
  static constexpr int NWL = (NWP + 511) / 512;         // per thread (the surplus ones repeat the thread's first piece)
  static constexpr int WBYTES = NWP * 16;
  static constexpr bool WDB = 2 * BUF + 2 * WBYTES + MT * 64 <= 160 * 1024;
  static constexpr int SMEM = 2 * BUF + (WDB ? 2 : 1) * WBYTES + MT * 64;
};

template <int MT, bool ST, int EPI>
__global__ __launch_bounds__(512, 1) void conv3d_split_fwd3_kernel(const SplitFwdArgs a) {
  using C = F3Cfg<MT>;
  constexpr int NSTEP = NSTEP27, RW = 2, NTHR = 512;
  constexpr int NITEM = RW * MT;  // epilogue items per lane: one f32x4 (4 channels of a voxel) each
  constexpr bool WDB = C::WDB;
  static_assert(!ST || EPI <= 1, "statistics belong to plain forward convs");
  static_assert(NITEM <= NSTEP, "one epilogue item per K step");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const wlds = lds + 2 * BUF;  // weight fragments: [buffer][piece 3][step 7][mt][lane] x 16 bytes
  float* const lbias = reinterpret_cast<float*>(wlds + (WDB ? 2 : 1) * C::WBYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g = lane >> 4, zw = wave & 3, yh = wave >> 2;
  const int chunk = blockIdx.y, nchunks = gridDim.y;
  const TileWalk walk = tile_walk(a.ntiles);
  const int tiles0 = a.ntiles / (a.tiles1 * a.tiles2);
  const int D0 = a.D0, D1 = a.D1, D2 = a.D2, Cin = a.Cin, Cout = a.Cout, ncc = a.ncc;

  int koff[NSTEP];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    int tap = syn_split_tap(4 * s + g);
    if (tap < 0) tap = syn_split_tap((4 * s + g) ^ 1);
    koff[s] = ((tap / 9 * HY + (tap / 3) % 3) * HX + tap % 3) * 16;
  }
  const int xv = syn_split_voxel(m);
  const int lbase = ((zw * HY + RW * yh) * HX + xv) * 16;

  // halo staging: 16-byte piece j = tid + 512 i (i < 3) of the 8-channel image -> halo voxel j >> 1, channels 4 (j & 1) .. + 3;
  // a thread without a third piece repeats its second one (same value to the same address: no divergent tail)
  constexpr int NP = HVOX * 2, NL = 3;
  static_assert(NL * NTHR >= NP && (NL - 1) * NTHR < NP, "three staging pieces per thread");
  int prel[NL], plds[NL];
  uint32_t pmask[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int j = tid + NTHR * i;
    if (j >= NP) j -= NTHR;
    const int v = j >> 1, h = j & 1;
    const int hz = v / (HY * HX), r = v - hz * (HY * HX), hy = r / HX, hx = r - hy * HX;
    prel[i] = ((hz * D1 + hy) * D2 + hx) * Cin * 4 + h * 16;
    plds[i] = v * 16 + h * 8;
    pmask[i] = (1u << hz) | (1u << (6 + hy)) | (1u << (12 + hx));
  }
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)((int64_t)D0 * D1 * D2 * Cin * 4), 0x00020000);
  // two staging sets: hst is converted during a chunk (it holds the NEXT chunk's image), hnx was requested at the end of the
  // chunk before and holds the image after that; at the end of a chunk hnx moves to hst (a chunk old: it has landed) and the
  // following image is requested -- every request has two chunk times to arrive
  f32x4 hst[NL], hnx[NL];
  // (tile, chunk) -> the one after it; past the end of the walk every lane reads out of range (zeros, never used)
  auto halo_issue = [&](int t, int cc) {
    const bool none = t >= walk.end;
    int z0, y0, x0;
    tile_decode(none ? walk.pos : t, tiles0, a.tiles1, a.tiles2, z0, y0, x0);
    uint32_t bad = none ? 0xFFFFFFFFu : 0x80000000u;
    bad |= syn_oob_bits(z0 - 1, HZ, D0) | (syn_oob_bits(y0 - 1, HY, D1) << 6) | (syn_oob_bits(x0 - 1, HX, D2) << 12);
    const int base = ((((z0 - 1) * D1 + (y0 - 1)) * D2 + (x0 - 1)) * Cin + cc * 8) * 4;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const uint32_t vo = (pmask[i] & bad) ? OOB : (uint32_t)(prel[i] + base);
#if SYN_ABL & 1
      hnx[i] = (f32x4){__uint_as_float(vo), 1.f, 2.f, 3.f};
#else
      hnx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (int)vo, 0, 0));
#endif
    }
  };
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  uint32_t cp[6];
#if SYN_ABL & 2
  auto conv_a = [&](int i) { cp[0] = __float_as_uint(hst[i][0]); cp[1] = __float_as_uint(hst[i][1]); cp[2] = cp[0] ^ cp[1]; };
  auto conv_b = [&](int i) { cp[3] = __float_as_uint(hst[i][2]); cp[4] = __float_as_uint(hst[i][3]); cp[5] = cp[3] ^ cp[4]; };
#else
  auto conv_a = [&](int i) { syn_split3(hst[i][0], hst[i][1], cp[0], cp[1], cp[2]); };
  auto conv_b = [&](int i) { syn_split3(hst[i][2], hst[i][3], cp[3], cp[4], cp[5]); };
#endif
  auto conv_c = [&](int i, unsigned char* dst) {
#if SYN_ABL & 4
    if (cp[0] == 0x12345u && cp[5] == 0x54321u)  // (never)
#endif
    {
    *reinterpret_cast<u32x2*>(dst + plds[i]) = (u32x2){cp[0], cp[3]};
    *reinterpret_cast<u32x2*>(dst + PLANE + plds[i]) = (u32x2){cp[1], cp[4]};
    *reinterpret_cast<u32x2*>(dst + 2 * PLANE + plds[i]) = (u32x2){cp[2], cp[5]};
    }
  };

  // weight staging: piece j = tid + 512 k of the chunk's 3 x (7 MT KB) fragment blocks
  const int piece_stride = nchunks * ncc * NSTEP * MT * 64;  // u32x4 elements between the bf16 pieces of the packed set
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u32x4*>(a.wp), 0, (int)((int64_t)3 * piece_stride * 16), 0x00020000);
  int wsrc[C::NWL];
#pragma unroll
  for (int k = 0; k < C::NWL; ++k) {
    int j = tid + NTHR * k;
    if (j >= C::NWP) j = tid;
    const int q = j / (NSTEP * MT * 64), r = j - q * (NSTEP * MT * 64);
    wsrc[k] = (q * piece_stride + chunk * ncc * NSTEP * MT * 64 + r) * 16;
  }
  u32x4 wst[C::NWL];
  bool wfirst = true;
  (void)wfirst;
  auto w_issue = [&](int cc) {
#if SYN_ABL & 8
    if (!wfirst) return;
#endif
#pragma unroll
    for (int k = 0; k < C::NWL; ++k) wst[k] = __builtin_amdgcn_raw_buffer_load_b128(rw, wsrc[k] + cc * (NSTEP * MT * 64 * 16), 0, 0);
  };
  auto w_store = [&](unsigned char* dst) {
#if SYN_ABL & 8
    if (!wfirst) return;
#endif
#pragma unroll
    for (int k = 0; k < C::NWL; ++k) {
      const int j = (tid + NTHR * k) >= C::NWP ? tid : tid + NTHR * k;
      *reinterpret_cast<u32x4*>(dst + j * 16) = wst[k];
    }
  };

  float s1[ST ? MT : 1][4], s2[ST ? MT : 1][4];
#pragma unroll
  for (int mt = 0; mt < (ST ? MT : 1); ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) s1[mt][i] = s2[mt][i] = 0.f;

  const int64_t out_bytes = (int64_t)D0 * D1 * D2 * Cout * 4;
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t radd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.addend ? a.addend : a.out), 0, (int)out_bytes, 0x00020000);
  const uint32_t ystep = (uint32_t)(D2 * Cout * 4);
  if (tid < MT * 16) {
    const int co = chunk * MT * 16 + tid;
    lbias[tid] = (a.bias && co < Cout) ? a.bias[co] : 0.f;
  }

  f32x4 acc[RW][MT], pend[RW][MT];
  uint32_t prow0 = OOB, pyok = 0;
  f32x4 ev, eb[EPI >= 2 ? NITEM : 1];  // eb: the parked tile's addend values, requested when the tile was parked
  uint32_t eoff;
  auto item_off = [&](int j, uint32_t row0, uint32_t yok) -> uint32_t {
    const int mt = j / RW, y = j % RW;
    const int co = (chunk * MT + mt) * 16 + 4 * g;
#if SYN_ABL & 32
    yok = 0;
#endif
    return (((yok >> y) & 1u) && co < Cout) ? row0 + (uint32_t)y * ystep + (uint32_t)(co * 4) : OOB;
  };
  auto addend_issue = [&](uint32_t row0, uint32_t yok) {
    if constexpr (EPI >= 2) {
#pragma unroll
      for (int j = 0; j < NITEM; ++j)
        eb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(radd, (int)item_off(j, row0, yok), 0, 0));
    }
  };
  auto epi0 = [&](int j, uint32_t row0, uint32_t yok) { eoff = item_off(j, row0, yok); };
  auto epi1 = [&](int j, const f32x4& src) {
    const f32x4 bj = *reinterpret_cast<const f32x4*>(lbias + (j / RW) * 16 + 4 * g);
    f32x4 v = src;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += bj[i];
    if constexpr (EPI == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] *= elu_dy(eb[EPI >= 2 ? j : 0][i]);
    }
    if constexpr (EPI == 3 || EPI == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += eb[EPI >= 2 ? j : 0][i];
    }
    if constexpr (EPI == 1 || EPI == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = elu_f(v[i]);
    }
    ev = v;
  };
  auto epi2 = [&](int j) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ev), rout, (int)eoff, 0, 0);
    if constexpr (ST) {
      const int mt = j / RW;
      const float w = eoff != OOB ? 1.f : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float r = w * ev[i];
        s1[mt][i] += r;
        s2[mt][i] += r * r;
      }
    }
  };
  auto tile_rows = [&](int t, uint32_t& row0, uint32_t& yok) {
    int z0, y0, x0;
    tile_decode(t, tiles0, a.tiles1, a.tiles2, z0, y0, x0);
    const int gz = z0 + zw, gx = x0 + xv, gy = y0 + RW * yh;
    const bool zx_ok = gz < D0 && gx < D2;
    row0 = (uint32_t)((gz * D1 + gy) * D2 + gx) * (uint32_t)(Cout * 4);
    yok = 0;
#pragma unroll
    for (int y = 0; y < RW; ++y) yok |= (zx_ok && (gy + y) < D1) ? (1u << y) : 0u;
  };

  u32x4 wa[2][3][MT], xb[2][3][RW];
  int buf = 0;
  if (walk.pos < walk.end) {  // prologue: image and weights of the first chunk, the second chunk's halo on its way
    halo_issue(walk.pos, 0);
    w_issue(0);
#pragma unroll
    for (int i = 0; i < NL; ++i) hst[i] = hnx[i];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      conv_a(i);
      conv_b(i);
      conv_c(i, lds);
    }
    w_store(wlds);
    wfirst = false;
    const int t1 = ncc > 1 ? walk.pos : walk.pos + walk.stride, c1 = ncc > 1 ? 1 : 0;
    halo_issue(t1, c1);
#pragma unroll
    for (int i = 0; i < NL; ++i) hst[i] = hnx[i];
    const bool last1 = c1 + 1 == ncc;
    halo_issue(last1 ? t1 + walk.stride : t1, last1 ? 0 : c1 + 1);
  }
  // one K chunk of tile t; WITH_EPI: the parked tile's epilogue rides along
  auto body = [&](int t, int cc, auto WITH_EPI) {
    constexpr bool WE = decltype(WITH_EPI)::value;
    __syncthreads();  // image `buf` and this chunk's weights are complete; nobody reads the other image any more
    const bool last_cc = cc + 1 == ncc;
    const int t1 = last_cc ? t + walk.stride : t, c1 = last_cc ? 0 : cc + 1;        // the next chunk ...
    const bool last1 = c1 + 1 == ncc;
    const int t2 = last1 ? t1 + walk.stride : t1, c2 = last1 ? 0 : c1 + 1;          // ... the one after it ...
    const bool last2 = c2 + 1 == ncc;
    const int t3 = last2 ? t2 + walk.stride : t2, c3 = last2 ? 0 : c2 + 1;          // ... and the third
    w_issue(c1);
    const unsigned char* img = lds + buf * BUF + lbase;
    unsigned char* nimg = lds + (buf ^ 1) * BUF;
    const unsigned char* wl = wlds + (WDB ? buf * C::WBYTES : 0) + lane * 16;
    auto frags = [&](auto SS, auto PP) {  // piece PP of the operands of step SS -> register set SS & 1
      constexpr int s = decltype(SS)::value, q = decltype(PP)::value;
#if SYN_ABL & 16
      if (s > 1) return;
#endif
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        wa[s & 1][q][mt] = *reinterpret_cast<const u32x4*>(wl + ((q * NSTEP + s) * MT + mt) * 1024);
#pragma unroll
      for (int y = 0; y < RW; ++y) xb[s & 1][q][y] = *reinterpret_cast<const u32x4*>(img + q * PLANE + koff[s] + y * (HX * 16));
    };
    auto mma = [&](auto SS, auto QA, auto QB) {
      constexpr int s = decltype(SS)::value, qa = decltype(QA)::value, qb = decltype(QB)::value;
#pragma unroll
      for (int y = 0; y < RW; ++y)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#if SYN_ABL & 64
          acc[y][mt][0] += __uint_as_float(wa[s & 1][qa][mt][0] ^ xb[s & 1][qb][y][0]);
#else
          acc[y][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[s & 1][qa][mt]),
                                                               __builtin_bit_cast(bf16x8, xb[s & 1][qb][y]), acc[y][mt], 0, 0, 0);
#endif
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    frags(I0{}, I2{});
    frags(I0{}, I1{});
    frags(I0{}, I0{});
    sfor<0, NSTEP>([&](auto S) {
      constexpr int s = decltype(S)::value;
      constexpr bool last = s + 1 == NSTEP;
      using SN = std::integral_constant<int, last ? 0 : s + 1>;
      constexpr int ci = s - (NSTEP - NL);                      // staging piece converted in this step (the last NL steps)
      constexpr bool cv = ci >= 0;
      constexpr bool ep = WE && s < NITEM;                      // epilogue item s of the parked tile
      __builtin_amdgcn_sched_barrier(0);
      mma(S, I0{}, I2{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!last) frags(SN{}, I2{});
      if constexpr (ep) epi0(s, prow0, pyok);
      if constexpr (cv) conv_a(ci);
      __builtin_amdgcn_sched_barrier(0);
      mma(S, I0{}, I1{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ep) epi1(s, pend[s % RW][s / RW]);
      __builtin_amdgcn_sched_barrier(0);
      mma(S, I0{}, I0{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!last) frags(SN{}, I1{});
      if constexpr (cv) conv_b(ci);
      __builtin_amdgcn_sched_barrier(0);
      mma(S, I1{}, I0{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ep) epi2(s);
      __builtin_amdgcn_sched_barrier(0);
      mma(S, I1{}, I1{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!last) frags(SN{}, I0{});
      if constexpr (cv) conv_c(ci, nimg);
      __builtin_amdgcn_sched_barrier(0);
      mma(S, I2{}, I0{});
    });
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NL; ++i) hst[i] = hnx[i];  // (t2, c2), requested a chunk ago
    halo_issue(t3, c3);
    if constexpr (!WDB) __syncthreads();  // single weight buffer: everyone has read its fragments
    w_store(wlds + (WDB ? (buf ^ 1) * C::WBYTES : 0));
    buf ^= 1;
  };
  auto zero_acc = [&]() {
#pragma unroll
    for (int y = 0; y < RW; ++y)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[y][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  // the first tile's rider works on an empty parked tile (pend = 0, every row invalid: nothing is stored or counted)
#pragma unroll
  for (int y = 0; y < RW; ++y)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) pend[y][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (walk.pos < walk.end) {
    for (int t = walk.pos; t < walk.end; t += walk.stride) {
      zero_acc();
      body(t, 0, std::true_type{});
      for (int cc = 1; cc < ncc; ++cc) body(t, cc, std::false_type{});
#pragma unroll
      for (int y = 0; y < RW; ++y)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) pend[y][mt] = acc[y][mt];
      tile_rows(t, prow0, pyok);
      addend_issue(prow0, pyok);
    }
#pragma unroll
    for (int j = 0; j < NITEM; ++j) {
      epi0(j, prow0, pyok);
      epi1(j, pend[j % RW][j / RW]);
      epi2(j);
    }
  }
  if constexpr (ST) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);  // [wave 8][2][MT*16]
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
    float* dst = a.stats_partial + (int64_t)blockIdx.x * (2 * Cout);
    for (int e = tid; e < MT * 16; e += NTHR) {
      const int c = chunk * MT * 16 + e;
      if (c < Cout) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          t1 += red[(w * 2 + 0) * (MT * 16) + e];
          t2 += red[(w * 2 + 1) * (MT * 16) + e];
        }
        dst[c] = t1;
        dst[Cout + c] = t2;
      }
    }
  }
}

// Forward pass of the up-sampled channel range of a folded decoder conv: out[2 v + p] (+)= sum over the 8 taps (a, b, c) of parity
// p's 2x2x2 window of W_p[(a, b, c)] . lo[v + (a, b, c) + p - 1], for all parities p from ONE staged (converted) low-resolution
// halo image -- a converted element feeds 8 parities x 8 taps instead of 27 taps.  512 threads: wave = (z plane, y half) of the
// 4x4x16 low-resolution tile, two x-rows per wave; accumulators for NPAR parities x 2 rows x MT co-tiles stay in registers over
// the input-channel chunks (register budget: NPAR = 4, or 2 for MT = 3; blockIdx.z = parity group).  Per chunk 2 NPAR K steps (parity, step) of
// 6 x 2 x MT MFMAs, weights one step ahead (two register sets), activations re-loaded per step.  D0..D2 = the low-res grid.
template <int MT, int NPAR, int NPROD = 6>
__global__ __launch_bounds__(512, 1) void conv3d_split_upfwd_kernel(const SplitFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // Round 6 (second half): wave = (parity q of the group, part of the tile) instead of (z plane, y half) x all parities: a wave
  // then streams ONLY its parity's weight fragments (the eight waves used to load the same 6 MT KB per K step each: 9 GB of L2
  // reads per 80^3 launch) and multiplies them into NR = 2 NPAR voxel rows -- 6 NR MT MFMAs per 3 MT weight fragments instead of
  // 12 MT.  Every accumulator receives the same products in the same order as before: bit-identical results.
  constexpr int NTHR = 512, WPP = 8 / NPAR, NR = 16 / WPP, NH = NR / 4;  // waves per parity, x-rows per wave, halves of 4 rows
  static_assert(NPAR == 2 || NPAR == 4, "two or four parities per workgroup");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4, q = wave % NPAR, part = wave / NPAR;
  const int pg = blockIdx.z;  // parity group: parities pg * NPAR + q
  const TileWalk walk = tile_walk(a.ntiles);
  const int tiles0 = a.ntiles / (a.tiles1 * a.tiles2);
  const int D0 = a.D0, D1 = a.D1, D2 = a.D2, Cin = a.Cin, Cout = a.Cout, ncc = a.ncc;

  int koff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int t8 = syn_split_tap8(4 * s + g);
    koff[s] = (((t8 >> 2) * HY + ((t8 >> 1) & 1)) * HX + (t8 & 1)) * 16;
  }
  const int xv = syn_split_voxel(m);
  // rows of this wave: r = part NR + j -> (z = r / 4, y = r % 4); window origin of its parity: halo offset p per axis
  const int par = pg * NPAR + q, pz = (par >> 2) & 1, py = (par >> 1) & 1, px = par & 1;
  const int zw0 = part * NH;
  const int lbase = ((zw0 * HY) * HX + xv) * 16 + ((pz * HY + py) * HX + px) * 16;

  constexpr int NP = HVOX * 2, NL = (NP + NTHR - 1) / NTHR;
  int prel[NL], plds[NL];
  uint32_t pmask[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int j = tid + NTHR * i;
    const int v = j >> 1, h = j & 1;
    const int hz = v / (HY * HX), r = v - hz * (HY * HX), hy = r / HX, hx = r - hy * HX;
    prel[i] = ((hz * D1 + hy) * D2 + hx) * Cin * 4 + h * 16;
    plds[i] = v * 16 + h * 8;
    pmask[i] = j < NP ? ((1u << hz) | (1u << (6 + hy)) | (1u << (12 + hx))) : 0xFFFFFFFFu;
  }
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)((int64_t)D0 * D1 * D2 * Cin * 4), 0x00020000);
  f32x4 stg[NL];
  auto load_halo = [&](int t, int cc) {
    int z0, y0, x0;
    tile_decode(t, tiles0, a.tiles1, a.tiles2, z0, y0, x0);
    uint32_t bad = 0x80000000u;
    bad |= syn_oob_bits(z0 - 1, HZ, D0) | (syn_oob_bits(y0 - 1, HY, D1) << 6) | (syn_oob_bits(x0 - 1, HX, D2) << 12);
    const int base = ((((z0 - 1) * D1 + (y0 - 1)) * D2 + (x0 - 1)) * Cin + cc * 8) * 4;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const uint32_t vo = (pmask[i] & bad) ? OOB : (uint32_t)(prel[i] + base);
      stg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (int)vo, 0, 0));
    }
  };
  auto store_halo = [&](int buf) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    unsigned char* dst = lds + buf * BUF;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      if (i == NL - 1 && tid + NTHR * i >= NP) continue;
      uint32_t p0, p1, p2, q0, q1, q2;
      syn_split3(stg[i][0], stg[i][1], p0, p1, p2);
      syn_split3(stg[i][2], stg[i][3], q0, q1, q2);
      *reinterpret_cast<u32x2*>(dst + plds[i]) = (u32x2){p0, q0};
      *reinterpret_cast<u32x2*>(dst + PLANE + plds[i]) = (u32x2){p1, q1};
      *reinterpret_cast<u32x2*>(dst + 2 * PLANE + plds[i]) = (u32x2){p2, q2};
    }
  };

  // weights: 8 parity sets [piece 3][cc][step 2][mt][lane] back to back (one co-chunk: Cout <= 16 MT); this wave's parity only
  const int64_t piece_stride = (int64_t)ncc * 2 * MT * 64, par_stride = 3 * piece_stride;
  const u32x4* __restrict__ wbase = a.wp + (int64_t)par * par_stride + lane;
  u32x4 wa[2][3][MT], xb[3][4];
  auto wload = [&](int cc, int s, int slot) {
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) wa[slot][pc][mt] = wbase[pc * piece_stride + ((cc * 2 + s) * MT + mt) * 64];
  };

  const int64_t out_bytes = (int64_t)8 * D0 * D1 * D2 * Cout * 4;
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t radd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.addend ? a.addend : a.out), 0, (int)out_bytes, 0x00020000);

  int buf = 0;
  if (walk.pos < walk.end) {
    load_halo(walk.pos, 0);
    store_halo(0);
    wload(0, 0, 0);
  }
  for (int t = walk.pos; t < walk.end; t += walk.stride) {
    f32x4 acc[NR][MT];
#pragma unroll
    for (int y = 0; y < NR; ++y)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[y][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int cc = 0; cc < ncc; ++cc) {
      __syncthreads();
      const bool more = cc + 1 < ncc || t + walk.stride < walk.end;
      if (cc + 1 < ncc) load_halo(t, cc + 1);
      else if (t + walk.stride < walk.end) load_halo(t + walk.stride, 0);
      const unsigned char* img = lds + buf * BUF + lbase;
      const int cc_next = cc + 1 < ncc ? cc + 1 : 0;
      sfor<0, 2>([&](auto S) {
        constexpr int s = decltype(S)::value, slot = s;
        sfor<0, NH>([&](auto H) {
          constexpr int h = decltype(H)::value;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int pc = 0; pc < 3; ++pc)
#pragma unroll
            for (int y = 0; y < 4; ++y)
              xb[pc][y] = *reinterpret_cast<const u32x4*>(img + pc * PLANE + koff[s] + (h * HY + y) * (HX * 16));
          if constexpr (h == 0) {  // the next step's weights (other register set), one step ahead
            if constexpr (s == 0) wload(cc, 1, 1);
            else wload(cc_next, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          sfor<0, NPROD>([&](auto CC) {
            constexpr int c = decltype(CC)::value;
            constexpr int qa = split_combo_a(c, NPROD), qb = split_combo_b(c, NPROD);
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
                acc[4 * h + y][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, wa[slot][qa][mt]), __builtin_bit_cast(bf16x8, xb[qb][y]), acc[4 * h + y][mt], 0, 0, 0);
          });
        });
      });
      __builtin_amdgcn_sched_barrier(0);
      if (more) store_halo(buf ^ 1);
      buf ^= 1;
    }
    // ---- epilogue: lane (m, g): channels mt*16 + 4g + i of voxel 2 (z0 + zw0 + r / 4, y0 + r % 4, x0 + xv) + parity
    int z0, y0, x0;
    tile_decode(t, tiles0, a.tiles1, a.tiles2, z0, y0, x0);
    const int gx = x0 + xv;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int co = mt * 16 + 4 * g;
      f32x4 bias = {0.f, 0.f, 0.f, 0.f};
      if (a.bias && co < Cout) bias = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int gz = z0 + zw0 + (r >> 2), gy = y0 + (r & 3);
        const bool vok = gz < D0 && gx < D2 && gy < D1 && co < Cout;
        const uint32_t off = vok ? (uint32_t)(((2 * gz + pz) * (2 * D1) + (2 * gy + py)) * (2 * D2) + (2 * gx + px)) *
                                       (uint32_t)(Cout * 4) + (uint32_t)(co * 4)
                                 : OOB;
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[r][mt][i] + bias[i];
        if (a.addend) {
          const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(radd, (int)off, 0, 0));
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += b[i];
        }
        if (a.act == 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = elu_f(v[i]);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rout, (int)off, 0, 0);
      }
    }
  }
}

// grid widths (workgroups along x; y / z = channel chunks / parity groups).  Never narrower than the tile count where that is
// possible: 296 workgroups for 300 tiles made 8 stragglers walk two tiles = twice the kernel's critical path.
inline int split_upfwd_grid_x(int ntiles, int ngroups) {
  int gx = std::max(8, ((256 / ngroups) / 8) * 8);  // one 512-thread workgroup per CU
  while (gx > 8 && gx - 8 >= ntiles) gx -= 8;
  if (ntiles < 8) gx = ntiles;
  return gx;
}
inline int split_wgrad_grid_x(int ntiles, int gy) {
  int gx = std::max(1, 256 / gy);  // one 512-thread workgroup per CU
  if (gx > ntiles) gx = ntiles;
  return gx;
}
inline int split_grid_x(int ntiles, int nchunks) {
  int gx = std::max(8, ((512 / nchunks) / 8) * 8);  // 2 workgroups per CU in total
  while (gx > 8 && gx - 8 >= ntiles) gx -= 8;
  if (ntiles < 8) gx = ntiles;
  return gx;
}

template <int MT, int NPAR, int NPROD>
int launch_split_upfwd_np(const SplitFwdArgs& a, hipStream_t st) {
  constexpr int NG = 8 / NPAR;
  const int gx = split_upfwd_grid_x(a.ntiles, NG);
  const size_t smem = 2 * BUF;
  auto kern = conv3d_split_upfwd_kernel<MT, NPAR, NPROD>;
  static SynOncePerDevice attr_done;
  if (auto once_ = attr_done.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  hipLaunchKernelGGL(kern, dim3(gx, 1, NG), dim3(512), smem, st, a);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

template <int MT, int NPAR>
int launch_split_upfwd(const SplitFwdArgs& a, hipStream_t st) {
  return t_nprod == 9 ? launch_split_upfwd_np<MT, NPAR, 9>(a, st) : launch_split_upfwd_np<MT, NPAR, 6>(a, st);
}

template <int MT, bool ST, int UPM, int NPROD>
int launch_split_fwd_np(const SplitFwdArgs& a, int gx, int nchunks, hipStream_t st) {
  const size_t smem = 2 * BUF;
  auto kern = conv3d_split_fwd_kernel<MT, ST, UPM, NPROD>;
  static SynOncePerDevice attr_done;
  if (auto once_ = attr_done.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  hipLaunchKernelGGL(kern, dim3(gx, nchunks), dim3(256), smem, st, a);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

// split-K halves (conv3d_split_fwd2_kernel<..., KS = 2>): the launch leaves CUs single-occupied anyway (at most one 4-wave
// workgroup per CU) and the chunk count is even
inline bool split_fwd2_uses_halves(int gx, int nchunks, int ncc) {
  return (int64_t)gx * nchunks <= 256 && ncc >= 4 && (ncc % 2) == 0;
}

template <int MT, bool ST, int EPI, bool STK>
int launch_split_fwd2_e(const SplitFwdArgs& a, int gx, int nchunks, hipStream_t st) {
  if constexpr (!STK) {
    if (split_fwd2_uses_halves(gx, nchunks, a.ncc)) {
      const size_t smem2 = 4 * BUF2 + MT * 16 * 4;
      auto kern2 = conv3d_split_fwd2_kernel<MT, ST, EPI, false, 2>;
      static SynOncePerDevice attr2_done;
      if (auto once_ = attr2_done.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
      }
      hipLaunchKernelGGL(kern2, dim3(gx, nchunks), dim3(512), smem2, st, a);
      return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
    }
  }
  const size_t smem = 2 * BUF2 + MT * 16 * 4;
  auto kern = conv3d_split_fwd2_kernel<MT, ST, EPI, STK>;
  static SynOncePerDevice attr_done;
  if (auto once_ = attr_done.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  hipLaunchKernelGGL(kern, dim3(gx, nchunks), dim3(256), smem, st, a);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

template <int MT, bool ST, bool STK = false>
int launch_split_fwd2(const SplitFwdArgs& a, int gx, int nchunks, hipStream_t st) {
  const int epi = a.act + ((a.addend && a.act < 2) ? 3 : 0);
  if constexpr (ST) {
    return epi == 1 ? launch_split_fwd2_e<MT, true, 1, STK>(a, gx, nchunks, st) : launch_split_fwd2_e<MT, true, 0, STK>(a, gx, nchunks, st);
  } else {
    switch (epi) {
      case 0: return launch_split_fwd2_e<MT, false, 0, STK>(a, gx, nchunks, st);
      case 1: return launch_split_fwd2_e<MT, false, 1, STK>(a, gx, nchunks, st);
      case 2: return launch_split_fwd2_e<MT, false, 2, STK>(a, gx, nchunks, st);
      case 3: return launch_split_fwd2_e<MT, false, 3, STK>(a, gx, nchunks, st);
      default: return launch_split_fwd2_e<MT, false, 4, STK>(a, gx, nchunks, st);
    }
  }
}

// which forward kernel (measured, profiles/r04_split_fwd_variants.txt): the 4-wave kernel with two workgroups per CU, except
// where the layer has between one and two rounds of its 512 workgroup slots AND several co-chunks (40^3, 96 output channels:
// 600 units): there the 8-wave kernel (one workgroup per CU, weights through LDS) quantises better (-10 %); with fewer units
// than slots (20^3, 192 output channels: 200) the 4-wave kernel wins again (0.126 vs 0.14 ms)
// Round 6: a layer that syn_split_plan_mt re-planned onto 32-channel co-chunks (40^3, 96 output channels: 900 units) stays on the
// 4-wave kernel -- 0.200 / 0.193 ms against 0.234 / 0.218 for the 8-wave kernel on 600 units, same box
// (profiles/r06_plan_40cubed_ab.txt)
inline bool split_uses_fwd3(int ntiles, int nchunks, int mt = 3, int Cout = 0) {
  const int64_t units = (int64_t)ntiles * nchunks;
  if (syn_split_replanned(ntiles, (Cout + 15) / 16, mt)) return false;
  return nchunks >= 2 && units >= 512 && units < 1024;
}

template <int MT, bool ST, int EPI>
int launch_split_fwd3_e(const SplitFwdArgs& a, int nchunks, hipStream_t st) {
  const int gx = split_upfwd_grid_x(a.ntiles, nchunks);  // one 512-thread workgroup per CU
  const size_t smem = F3Cfg<MT>::SMEM;
  auto kern = conv3d_split_fwd3_kernel<MT, ST, EPI>;
  static SynOncePerDevice attr_done;
  if (auto once_ = attr_done.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  hipLaunchKernelGGL(kern, dim3(gx, nchunks), dim3(512), smem, st, a);
  return hipGetLastError() == hipSuccess ? SYNTHSR_OK : SYNTHSR_ELAUNCH;
}

template <int MT, bool ST>
int launch_split_fwd3(const SplitFwdArgs& a, int nchunks, hipStream_t st) {
  const int epi = a.act + ((a.addend && a.act < 2) ? 3 : 0);
  if constexpr (ST) {
    return epi == 1 ? launch_split_fwd3_e<MT, true, 1>(a, nchunks, st) : launch_split_fwd3_e<MT, true, 0>(a, nchunks, st);
  } else {
    switch (epi) {
      case 0: return launch_split_fwd3_e<MT, false, 0>(a, nchunks, st);
      case 1: return launch_split_fwd3_e<MT, false, 1>(a, nchunks, st);
      case 2: return launch_split_fwd3_e<MT, false, 2>(a, nchunks, st);
      case 3: return launch_split_fwd3_e<MT, false, 3>(a, nchunks, st);
      default: return launch_split_fwd3_e<MT, false, 4>(a, nchunks, st);
    }
  }
}

template <int MT, bool ST, int UPM = 0>
int launch_split_fwd(const SplitFwdArgs& a, int gx, int nchunks, hipStream_t st) {
  if (a.stacked) {  // the layout decides the kernel (conv3d.hip plans it only for 6 products, Cout = 24, one co-chunk)
    if constexpr (UPM == 0 && MT == 2) return launch_split_fwd2<2, ST, true>(a, gx, nchunks, st);
    return SYNTHSR_EINVAL;
  }
  if constexpr (UPM == 0) {
    if (t_nprod == 6 && split_uses_fwd3(a.ntiles, nchunks, MT, a.Cout)) return launch_split_fwd3<MT, ST>(a, nchunks, st);
    if (t_nprod == 6) return launch_split_fwd2<MT, ST>(a, gx, nchunks, st);
  }
  return t_nprod == 9 ? launch_split_fwd_np<MT, ST, UPM, 9>(a, gx, nchunks, st) : launch_split_fwd_np<MT, ST, UPM, 6>(a, gx, nchunks, st);
}



// ------------------------------------------------------------------------------------------------ weight gradient
//   dW[tap][ci][co] = sum over voxels v of x[v + tap - 1][ci] * dz[v][co]:  D[(tap, ci)][co] += sum over the 6 (i, j) of
//   A_i[(tap, ci)][voxel] * B_j[voxel][co], both operands activations: both go through LDS as three bf16 pieces in their
//   natural [voxel][channel] layout and are read K(voxel)-major with the transpose read ds_read_b64_tr_b16 (as conv_bf16.hip's
//   weight-gradient kernel).  LDS bandwidth is what bounds this kernel (an operand fragment is 1 KB = 8 LDS cycles of a CU,
//   an MFMA 16 cycles of a SIMD), so the decomposition maximises fragment re-use:
//     * a workgroup owns 8 input channels (blockIdx.y) x COW output channels and ALL 27 taps = 14 row tiles of 16 rows =
//       (2 taps) x (2 channel quads) x 4 channels;
//     * COW = 24: its 8 waves split the tile's 256 voxels (one 32-voxel K step each) and every wave holds all 14 row tiles;
//       COW = 48 (3 column tiles: 14 x 3 accumulator tiles do not fit a wave): waves 0-3 hold row tiles 0-6, waves 4-7 row
//       tiles 7-13, each wave two K steps -- in both cases a wave loads the 3 x NT dz fragments of a K step once and streams
//       the 3 x RT x fragments past them: 6 NT MFMAs per 3 KB of LDS reads, and a staged (converted) tile serves all 27 taps;
//     * accumulators stay in registers over the workgroup's whole tile range; at the end the 8 waves' partial sums are added
//       through LDS and flushed with one atomic per element (deterministic mode: private planes, common.h DetRun).
//   8 waves x 1 workgroup per CU; LDS double buffered when it fits (COW = 24: 2 x 68 KB): one barrier per tile.
//   CIW = 16 (COW = 48 and Cin % 16 == 0, six products): 16 input channels per workgroup -- 28 row tiles of one tap x 16 channels in
//   four row groups of two waves, four K steps per wave, one 136 KB image.  The dz tile is loaded and converted once per 16
//   instead of once per 8 input channels and a barrier pair covers twice the matrix work: 8-21 % faster from 80^3 48->48 down
//   to 20^3 192->192 (profiles/r04_split_wgrad_ciw16_ab.txt).  The bias row of ones takes the spare 28th row tile.
//   CIW = 24 (the stacked 24-column kernel when Cin = 24: the 160^3 layers): all input channels in one workgroup, 40.5 row
//   tiles whose 4 channel quads straddle taps, four row groups of 11; dz converted once instead of three times per tile,
//   HBM traffic halves: 0.83 -> 0.72 ms inside the training step (profiles/r04_split_wgrad_ciw24_ab.txt).
//   STK (COW = 24, six products): 24 columns are 1.5 column tiles, and two padded tiles per dz piece spend a quarter of the
//   matrix instructions on columns nobody reads.  The transpose read takes an address per lane, so the column tiles are
//   STACKED purely by addressing (the LDS image is unchanged): U0, U1, U2 = channels 0-15 of pieces 0, 1, 2; U3 = channels
//   16-23 of piece 0 | of piece 1 (lane quads 0-1 | 2-3); U4 = channels 16-23 of piece 2 | zeros (the quads 2-3 read 8 zeroed
//   bytes behind the image).  x0 (U0 U1 U2 U3 U4), x1 (U0 U1 U3), x2 (U0 U3) = 10 MFMAs per row tile instead of 12 cover the
//   six products (plus x2 dz1 on channels 16-23, a term the six-product sum merely omits); columns 16-23 are accumulator
//   columns 0-7 + 8-15 of the second tile, added once before the flush.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__device__ __forceinline__ u32x4 tr_read8(const unsigned char* p_lo, const unsigned char* p_hi) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p_lo);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p_hi);
  return __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

struct SplitWgArgs {
  const float* in;    // x [vox][Cin]
  const float* dout;  // dz [vox][Cout]
  float* dw;          // [27][cin_total][Cout], accumulated with atomics
  float* dbias;       // [Cout] += sum over voxels of dz, or null
  int D0, D1, D2, Cin, Cout, cin_total, ci_off, ncc, nco, tiles1, tiles2, ntiles, ciw;
  int64_t det_stride;
};

template <int COW, int CIW = 8>
struct WgCfg {
  static constexpr int NT = (COW + 15) / 16, DROWB = COW * 2, DPLANE = TZ * TY * TX * DROWB;
  static constexpr int XB = CIW * 2, XPLANE = HVOX * XB;        // bytes of a halo voxel / of a piece plane of the x image
  // row tiles of 27 taps x CIW input channels (CIW = 8, 16: one spare half tile / tile; 24: 40.5 -> four groups of 11) ...
  static constexpr int NRT = CIW == 24 ? 44 : 14 * (CIW / 8);
  static constexpr int RT = CIW == 24 ? 11 : (COW <= 24 ? 14 : 7);  // ... and per wave
  static constexpr int BUFB = 3 * XPLANE + 3 * DPLANE + 64;     // + slack: the last column tile reads past a 24-channel row
  static constexpr bool DBUF = 2 * BUFB <= 160 * 1024;
  static constexpr int NBUF = DBUF ? 2 : 1;
};

// CIW = 24 (round 5; profiles/r05_split_wgrad_var24_ab.txt: 0.84 -> 0.69 ms on the same box, bit-identical results):
//   PL   the x image is stored as six channel-quad PLANES per piece ([quad][halo voxel] x 8 bytes) instead of voxel rows of 48
//        bytes: the 32 lanes a ds_read_b64_tr_b16 serves together (8 voxels x 4 quads) then read 4 x 64 contiguous bytes whose
//        plane offsets (5184 = 20 x 256 + 64 bytes) put them on four disjoint sets of 16 banks; with 48-byte rows voxel v + 5
//        landed on the banks of voxel v (a 2-way conflict on EVERY A read: SQ_LDS_BANK_CONFLICT was 45 % of the kernel's LDS
//        cycles, profiles/r04_pmc_lds_valu.txt).  Row tiles that straddle two taps (quads 4, 5 | 0, 1) keep a conflict where
//        the taps are x neighbours: 8 of the 41 tiles.
//   BAL  41 row tiles + the ones row = 42 slots in groups of 11, 10, 10, 11 instead of 4 x 11: waves w and w + 4 share a SIMD,
//        i.e. groups (0, 2) and (1, 3): 21 instead of 22 slots per SIMD and tile.
//   (Issuing the six reads of row tile q + 1 one by one behind the MFMAs of row tile q instead of as a burst: -2 % without BAL,
//   nothing with it -- not kept.)
template <int COW, int NPROD = 6, bool STK = false, int CIW = 8>
__global__ __launch_bounds__(512, 1) void conv3d_split_wgrad_kernel(const SplitWgArgs a) {
  constexpr bool PL = CIW == 24, BAL = CIW == 24;
  constexpr int QP = HVOX * 8;  // PL: bytes of one channel-quad plane of a piece
  static_assert(!STK || (COW == 24 && NPROD == 6), "stacked column tiles: 24 columns, six products");
  static_assert(CIW == 8 || (CIW == 16 && COW == 48) || (CIW == 24 && COW == 24 && STK),
                "16 input channels per workgroup: the 48-column kernel; 24: the stacked 24-column kernel");
  // CIW = 24 (Cin = 24, the 160^3 layers): ALL input channels in one workgroup -- 648 rows = 40.5 row tiles of 4 channel quads
  // (a tile straddles taps: quad Q = 4 tile + lq -> tap Q / 6, channels 4 (Q % 6) ..), four row groups of 11, two waves x four
  // K steps each, one 130 KB image: the dz tile is loaded and converted once instead of three times per tile.
  // CIW = 16 (Cin % 16 == 0): a workgroup owns 16 input channels = 28 row tiles (one tap x 16 channels each) in four row
  // groups of two waves with four K steps each: the dz tile is staged and converted once per 16 instead of once per 8 input
  // channels and a barrier pair covers twice the matrix work (one 136 KB image).
  using C = WgCfg<COW, CIW>;
  constexpr int NT = C::NT, DROWB = C::DROWB, DPLANE = C::DPLANE, RT = C::RT, NW = 8, NTHR = 512;
  constexpr int XB = C::XB, WG_XPLANE = C::XPLANE, XQ = CIW / 4;           // channel quads of a halo voxel
  constexpr int NXP = HVOX * XQ, NXL = (NXP + NTHR - 1) / NTHR;            // 16-byte pieces (4 channels) of the x halo image
  constexpr int DQ = COW / 4, NDP = TZ * TY * TX * DQ, NDL = NDP / NTHR;   // 16-byte pieces (4 channels) of the dz tile
  static_assert(NDP % NTHR == 0, "dz pieces divide among the threads");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15, lrow = li >> 2, lq = li & 3;
  constexpr int GROUPS = C::NRT / RT, WPG = NW / GROUPS, KPW = 8 / WPG;  // row groups, waves per group, K steps per wave
  // this wave's row group and its place in it (CIW = 24: as a scalar -- the address tables of 11 row tiles then cost no vector
  // registers; the other variants keep the round-4 form: their register allocation spills MORE with the scalar, measured)
  const int rh = CIW == 24 ? __builtin_amdgcn_readfirstlane(wave / WPG) : wave / WPG, wg = wave % WPG;
  // first row tile of the group and whether it has an 11th slot (BAL: 11 | 10 | 10 | 11 slots, the last one the ones row)
  const int rt0 = BAL ? rh * (RT - 1) + (rh > 0 ? 1 : 0) : rh * RT;
  const bool slot10 = !BAL || rh == 0 || rh == GROUPS - 1;
  const int cc = blockIdx.y % a.ncc, oc = blockIdx.y / a.ncc;
  const TileWalk walk = tile_walk(a.ntiles);
  const int tiles0 = a.ntiles / (a.tiles1 * a.tiles2);
  const int D0 = a.D0, D1 = a.D1, D2 = a.D2, Cin = a.Cin, Cout = a.Cout;

  // A addresses: lane i = (voxel row lrow, block lq) of the row tile: tap = 2 (rh RT + q) + (lq >> 1), channel quad lq & 1
  // (CIW = 16: tap = rh RT + q, channel quad lq; CIW = 24: quad Q = 4 (rh RT + q) + lq of the 27 x 6 quads)
  int aoff[RT];
#pragma unroll
  for (int q = 0; q < RT; ++q) {
    const int Q = 4 * (rt0 + q) + lq;
    int tap = CIW == 8 ? 2 * (rh * RT + q) + (lq >> 1) : (CIW == 16 ? rh * RT + q : Q / 6);
    if (tap > 26) tap = 26;  // the spare slot of the last pair / the spare tiles: their rows are not flushed
    const int tvox = (tap / 9 * HY + (tap / 3) % 3) * HX + tap % 3;
    if constexpr (PL) aoff[q] = tvox * 8 + (Q % 6) * QP;
    else aoff[q] = tvox * XB + (CIW == 8 ? (lq & 1) : (CIW == 16 ? lq : Q % 6)) * 8;
  }
  // this wave's K step: 32 voxels = x-rows (z, yb) and (z, yb + 1); K index 8 g + j <-> voxel (row g >> 1, x = 8 (j >> 2) +
  // 4 (g & 1) + (j & 3)) -- the same bijection for both operands (conv_bf16.hip)
  // (KPW = 2: K steps 2 wg and 2 wg + 1 = the four x-rows of z plane wg; the second one sits 2 rows further down)
  // (KPW = 4: the K steps of z planes 2 wg and 2 wg + 1)
  const int ks0 = wg * KPW, kz = ks0 >> 1, kyb = 2 * (ks0 & 1);
  // address of K step kj of this wave relative to its first one: two x-rows down, or (every second step) one z plane on
  constexpr int XVB = PL ? 8 : XB;  // bytes from a halo voxel to the next one (same channel quad)
  auto akoff = [](int kj) { return (uint32_t)(((kj >> 1) * HY * HX + (kj & 1) * 2 * HX) * XVB); };
  auto bkoff = [](int kj) { return (uint32_t)(((kj >> 1) * TY * TX + (kj & 1) * 2 * TX) * DROWB); };
  static_assert(KPW == 1 || KPW == 2 || KPW == 4, "K steps of a wave: whole x-row pairs of consecutive z planes");
  const int vx = 4 * (g & 1) + lrow, vr = g >> 1;
  const uint32_t abase = (uint32_t)((((kz * HY + kyb + vr) * HX) + vx) * XVB);
  const uint32_t bbase = (uint32_t)(3 * WG_XPLANE) + (uint32_t)((((kz * TY + kyb + vr) * TX) + vx) * DROWB + lq * 8);
  // stacked column tiles U3 / U4: channels 16-23 (byte 32 of the row) of piece lq >> 1 resp. of piece 2 | the zeroed slack
  constexpr uint32_t ZOFF = (uint32_t)(3 * WG_XPLANE + 3 * DPLANE);
  const uint32_t bvox = bbase - (uint32_t)(lq * 8);
  const uint32_t u3off = bvox + (uint32_t)((lq >> 1) * DPLANE + 32 + (lq & 1) * 8);
  const uint32_t u4off = lq < 2 ? bvox + (uint32_t)(2 * DPLANE + 32 + (lq & 1) * 8) : ZOFF;
  const uint32_t u4hi = lq < 2 ? (uint32_t)(8 * DROWB) : 0u;
  auto u4koff = [&](int kj) { return lq < 2 ? bkoff(kj) : 0u; };  // (the zero lanes take no K-step offset)
  if constexpr (STK) {
    if (tid < 2 * C::NBUF) *reinterpret_cast<uint64_t*>(lds + (tid >> 1) * C::BUFB + ZOFF + (tid & 1) * 8) = 0ull;
  }

  // staging: x piece j -> halo voxel j / XQ, channels 4 (j % XQ) .. + 3 of the chunk; dz piece j -> voxel j / DQ, quad j % DQ
  // (LDS address of piece j, either image: 8 j -- a voxel's XQ resp. DQ pieces are 8 bytes apart and fill its XB resp. DROWB bytes)
  static_assert(XB == XQ * 8 && DROWB == DQ * 8, "pieces tile the voxel rows");
  // XPLN (CIW = 16, round 6): piece i of a thread = halo PLANE i -- thread = (channel quad tid & 3, in-plane voxel tid >> 2 of the
  // 6 x 18 = 108), so its six pieces differ by one plane stride and one mask bit: two registers instead of twelve address / mask
  // registers (the 48-column kernel had 256 + 5 spilled); the same 16-byte pieces as before, 108 of 128 lanes active per load
  // where the linear mapping had 2592 of 3072
  constexpr bool XPLN = CIW == 16;
  static_assert(!XPLN || (NXL == HZ && NTHR / XQ >= HY * HX), "one halo plane per staging piece");
  int xrel[XPLN ? 1 : NXL];
  uint32_t xmask[XPLN ? 1 : NXL];
  const int xq = tid & 3, xr = tid >> 2;            // XPLN: channel quad, voxel inside a halo plane
  const int xps = D1 * D2 * Cin * 4;                // XPLN: bytes from a halo plane to the next
  if constexpr (XPLN) {
    const int hy = xr / HX, hx = xr - hy * HX;
    xrel[0] = (hy * D2 + hx) * Cin * 4 + xq * 16;
    xmask[0] = xr < HY * HX ? ((1u << (6 + hy)) | (1u << (12 + hx))) : 0xFFFFFFFFu;
  } else {
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
      const int j = tid + NTHR * i;
      const int v = j / XQ, h = j % XQ;
      const int hz = v / (HY * HX), r = v - hz * (HY * HX), hy = r / HX, hx = r - hy * HX;
      xrel[i] = ((hz * D1 + hy) * D2 + hx) * Cin * 4 + h * 16;
      xmask[i] = j < NXP ? ((1u << hz) | (1u << (6 + hy)) | (1u << (12 + hx))) : 0xFFFFFFFFu;
    }
  }
  // (XPLN, dz side: thread = (channel quad tid & 3 of four, voxel tid >> 2 of the first two z planes); piece i = channel quads
  //  4 (i % 3) .. of z plane pair i / 3: again one address and one mask per thread)
  constexpr bool DPLN = XPLN && COW == 48;
  int drel[DPLN ? 1 : NDL];
  uint32_t dmask[DPLN ? 1 : NDL];  // bits (z | 4 + y | 8 + x) of the voxel inside the tile, against the tile's out-of-volume bits
  const int dps = 2 * D1 * D2 * Cout * 4;           // DPLN: bytes from z plane pair 0 to pair 1
  if constexpr (DPLN) {
    static_assert(!DPLN || NDL == 6, "two plane pairs x three quad groups");
    const int vz = xr / (TY * TX), vy = (xr / TX) % TY, vxx = xr % TX;   // xr < 128: z planes 0, 1
    dmask[0] = (1u << vz) | (1u << (4 + vy)) | (1u << (8 + vxx));         // (COW divides Cout: every column exists)
    drel[0] = ((vz * D1 + vy) * D2 + vxx) * Cout * 4 + (oc * COW + xq * 4) * 4;
  } else {
#pragma unroll
    for (int i = 0; i < NDL; ++i) {
      const int j = tid + NTHR * i;
      const int v = j / DQ, c4 = j - v * DQ;
      const int vz = v / (TY * TX), vy = (v / TX) % TY, vxx = v % TX;
      const int co = oc * COW + c4 * 4;
      dmask[i] = co < Cout ? ((1u << vz) | (1u << (4 + vy)) | (1u << (8 + vxx))) : 0xFFFFFFFFu;
      drel[i] = ((vz * D1 + vy) * D2 + vxx) * Cout * 4 + co * 4;
    }
  }
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)((int64_t)D0 * D1 * D2 * Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dout), 0, (int)((int64_t)D0 * D1 * D2 * Cout * 4), 0x00020000);
  f32x4 xst[NXL], dst[NDL];
  auto load_tile = [&](int t) {
    int z0, y0, x0;
    tile_decode(t, tiles0, a.tiles1, a.tiles2, z0, y0, x0);
    uint32_t bad = 0x80000000u;
    bad |= syn_oob_bits(z0 - 1, HZ, D0) | (syn_oob_bits(y0 - 1, HY, D1) << 6) | (syn_oob_bits(x0 - 1, HX, D2) << 12);
    const int base = ((((z0 - 1) * D1 + (y0 - 1)) * D2 + (x0 - 1)) * Cin + cc * CIW) * 4;
    int xps_t = xps;
    uint32_t xm_t = xmask[0];
    if constexpr (XPLN) {  // keep the six addresses / masks from being hoisted out of the tile loop into twelve registers again
      asm volatile("" : "+s"(xps_t));
      asm volatile("" : "+v"(xm_t));
    }
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
      uint32_t vo;
      if constexpr (XPLN) vo = ((xm_t | (1u << i)) & bad) ? OOB : (uint32_t)(xrel[0] + i * xps_t + base);
      else vo = (xmask[i] & bad) ? OOB : (uint32_t)(xrel[i] + base);
      xst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (int)vo, 0, 0));
    }
    const int dbase = ((z0 * D1 + y0) * D2 + x0) * Cout * 4;
    uint32_t dbad = 0x80000000u;
    dbad |= syn_oob_bits(z0, TZ, D0) | (syn_oob_bits(y0, TY, D1) << 4) | (syn_oob_bits(x0, TX, D2) << 8);
    int dps_t = dps;
    uint32_t dm_t = dmask[0];
    if constexpr (DPLN) {
      asm volatile("" : "+s"(dps_t));
      asm volatile("" : "+v"(dm_t));
    }
#pragma unroll
    for (int i = 0; i < NDL; ++i) {
      int dvo;
      if constexpr (DPLN) {
        const uint32_t m = (dm_t & ~0xFu) | ((dm_t & 0xFu) << (2 * (i / 3)));
        dvo = (m & dbad) ? (int)OOB : drel[0] + (i / 3) * dps_t + (i % 3) * 64 + dbase;
      } else {
        dvo = (dmask[i] & dbad) ? (int)OOB : drel[i] + dbase;
      }
      dst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdo, dvo, 0, 0));
    }
  };
  auto store_tile = [&](int buf) {  // four fp32 -> 3 x (four bf16 = 8 bytes)
    unsigned char* xd = lds + buf * C::BUFB;
    unsigned char* dd = xd + 3 * WG_XPLANE;
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
      if constexpr (XPLN) {
        if (xr >= HY * HX) continue;
      } else {
        if (NXP % NTHR != 0 && i == NXL - 1 && tid + NTHR * i >= NXP) continue;
      }
      uint32_t p0, p1, p2, q0, q1, q2;
      syn_split3(xst[i][0], xst[i][1], p0, p1, p2);
      syn_split3(xst[i][2], xst[i][3], q0, q1, q2);
      const int xj = XPLN ? (i * (HY * HX) + xr) * XQ + xq : tid + NTHR * i;
      const int xl = PL ? (xj % XQ) * QP + (xj / XQ) * 8 : xj * 8;
      *reinterpret_cast<u32x2*>(xd + xl) = (u32x2){p0, q0};
      *reinterpret_cast<u32x2*>(xd + WG_XPLANE + xl) = (u32x2){p1, q1};
      *reinterpret_cast<u32x2*>(xd + 2 * WG_XPLANE + xl) = (u32x2){p2, q2};
    }
#pragma unroll
    for (int i = 0; i < NDL; ++i) {
      uint32_t p0, p1, p2, q0, q1, q2;
      syn_split3(dst[i][0], dst[i][1], p0, p1, p2);
      syn_split3(dst[i][2], dst[i][3], q0, q1, q2);
      const int dl = DPLN ? ((xr + 128 * (i / 3)) * DQ + xq + 4 * (i % 3)) * 8 : (tid + NTHR * i) * 8;
      *reinterpret_cast<u32x2*>(dd + dl) = (u32x2){p0, q0};
      *reinterpret_cast<u32x2*>(dd + DPLANE + dl) = (u32x2){p1, q1};
      *reinterpret_cast<u32x2*>(dd + 2 * DPLANE + dl) = (u32x2){p2, q2};
    }
  };

  // row tile RT: the bias gradient = (a row of ones) x dz, in the workgroups of the first input-channel chunk / row half only
  // (CIW = 16: no extra tile -- the ones sit in the spare 28th row tile, the last one of the last row group)
  constexpr int BT = CIW != 8 ? RT - 1 : RT, NACC = BT + 1;
  const bool want_db = a.dbias != nullptr && cc == 0 && rh == (CIW != 8 ? GROUPS - 1 : 0);
  const uint32_t one2 = li == 0 ? 0x3f803f80u : 0u;  // A fragment whose row 0 is all ones (bf16 1.0), exact in piece 0
  const u32x4 ones = {one2, one2, one2, one2};
  f32x4 acc[NACC][NT];
#pragma unroll
  for (int q = 0; q < NACC; ++q)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[q][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int buf = 0;
  if (walk.pos < walk.end) {
    load_tile(walk.pos);
    store_tile(0);
  }
  for (int t = walk.pos; t < walk.end; t += walk.stride) {
    __syncthreads();  // image `buf` is complete (double buffered: and nobody reads the other one any more)
    const bool more = t + walk.stride < walk.end;
    if (more) load_tile(t + walk.stride);
    const unsigned char* img = lds + buf * C::BUFB;
    sfor<0, KPW>([&](auto KJ) {
      constexpr int kj = decltype(KJ)::value;
      u32x4 bfr[3][NT], afr[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int n = 0; n < (STK ? 1 : NT); ++n)
          bfr[p][n] = tr_read8(img + bbase + bkoff(kj) + p * DPLANE + n * 32, img + bbase + bkoff(kj) + p * DPLANE + n * 32 + 8 * DROWB);
      if constexpr (STK) {  // bfr[0][1] = U3, bfr[1][1] = U4 (bfr[2][1] unused)
        bfr[0][1] = tr_read8(img + u3off + bkoff(kj), img + u3off + bkoff(kj) + 8 * DROWB);
        bfr[1][1] = tr_read8(img + u4off + u4koff(kj), img + u4off + u4koff(kj) + u4hi);
      }
      auto aload = [&](int q, int slot) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          afr[slot][p] = tr_read8(img + abase + akoff(kj) + aoff[q] + p * WG_XPLANE,
                                  img + abase + akoff(kj) + aoff[q] + p * WG_XPLANE + 8 * XVB);
          if (CIW != 8 && q == BT && want_db) afr[slot][p] = p == 0 ? ones : (u32x4){0u, 0u, 0u, 0u};
        }
      };
      aload(0, 0);
      if constexpr (CIW == 8) {
        if (want_db) {
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              if (STK && n == 1 && p == 2) continue;
              acc[BT][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones),
                                                                    __builtin_bit_cast(bf16x8, bfr[p][n]), acc[BT][n], 0, 0, 0);
            }
        }
      }
      sfor<0, RT>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        if (BAL && q == RT - 1 && !slot10) return;  // (wave-uniform) the 10-slot groups
        const bool pre = q + 1 < RT && (!BAL || q + 2 < RT || slot10);  // is there a next row tile to fetch
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (q + 1 < RT) {
          if (pre) aload(q + 1, (q + 1) & 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (STK) {
          // (x piece, column tile, accumulator): the two accumulators alternate; smallest terms first as in the plain order
          auto mm = [&](int xa, int pb, int nb, int ac) {
            acc[q][ac] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afr[q & 1][xa]),
                                                                  __builtin_bit_cast(bf16x8, bfr[pb][nb]), acc[q][ac], 0, 0, 0);
          };
          mm(0, 2, 0, 0);  // x0 U2
          mm(0, 1, 1, 1);  // x0 U4
          mm(2, 0, 0, 0);  // x2 U0
          mm(2, 0, 1, 1);  // x2 U3
          mm(1, 1, 0, 0);  // x1 U1
          mm(1, 0, 1, 1);  // x1 U3
          mm(0, 1, 0, 0);  // x0 U1
          mm(0, 0, 1, 1);  // x0 U3
          mm(1, 0, 0, 0);  // x1 U0
          mm(0, 0, 0, 0);  // x0 U0
        } else {
          sfor<0, NPROD>([&](auto CC) {
            constexpr int c = decltype(CC)::value;
            constexpr int qa = split_combo_a(c, NPROD), qb = split_combo_b(c, NPROD);
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[q][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afr[q & 1][qa]),
                                                                   __builtin_bit_cast(bf16x8, bfr[qb][n]), acc[q][n], 0, 0, 0);
          });
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (!C::DBUF) __syncthreads();  // single buffer: everyone is done reading before the next image is written
    if (more) store_tile(C::DBUF ? buf ^ 1 : 0);
    if constexpr (C::DBUF) buf ^= 1;
  }
  // ---- add the partial sums of the WPG waves of a row group through LDS (halving), the group's first wave flushes
  float* red = reinterpret_cast<float*>(lds);  // [row group][wave slot][RT + 1][NT][4][64]
  constexpr int WSZ = NACC * NT * 4 * 64;
  static_assert((size_t)GROUPS * (WPG / 2) * WSZ * 4 <= (size_t)C::NBUF * C::BUFB, "the reduction slots fit the LDS images");
#pragma unroll
  for (int half = WPG / 2; half >= 1; half >>= 1) {
    __syncthreads();
    if (wg >= half && wg < 2 * half) {
      float* dstw = red + (size_t)(rh * (WPG / 2) + wg - half) * WSZ;
#pragma unroll
      for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int i = 0; i < 4; ++i) dstw[((q * NT + n) * 4 + i) * 64 + lane] = acc[q][n][i];
    }
    __syncthreads();
    if (wg < half) {
      const float* srcw = red + (size_t)(rh * (WPG / 2) + wg) * WSZ;
#pragma unroll
      for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[q][n][i] += srcw[((q * NT + n) * 4 + i) * 64 + lane];
    }
  }
  if (wg != 0) return;
  if constexpr (STK) {  // columns 16-23 = columns 0-7 + 8-15 of the stacked tile (lane li + 8 of the same 16-lane row group)
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[q][1][i] += __shfl_down(acc[q][1][i], 8, 16);
  }
  // lane (li -> co, rows 4 g + i -> block g of the row tile: tap 2 (rh RT + q) + (g >> 1), channel 4 (g & 1) + i of the chunk)
  float* dwp = a.dw + (size_t)blockIdx.x * a.det_stride;
  if (want_db && g == 0) {  // row 0 of the ones tile
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int col = n * 16 + li, co = oc * COW + col;
      if (col < COW && co < Cout) atomicAdd(a.dbias + (size_t)blockIdx.x * a.det_stride + co, acc[BT][n][0]);
    }
  }
#pragma unroll
  for (int q = 0; q < RT; ++q) {
    if (BAL && q == RT - 1 && !slot10) continue;
    const int Qf = 4 * (rt0 + q) + g;  // CIW = 24: the channel quad of accumulator rows 4 g .. 4 g + 3
    const int tap = CIW == 8 ? 2 * (rh * RT + q) + (g >> 1) : (CIW == 16 ? rh * RT + q : Qf / 6);
    if (tap > 26) continue;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int col = n * 16 + li, co = oc * COW + col;
      if (col >= COW || co >= Cout) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ci = a.ci_off + cc * CIW + (CIW == 8 ? 4 * (g & 1) : (CIW == 16 ? 4 * g : 4 * (Qf % 6))) + i;
        atomicAdd(dwp + ((int64_t)tap * a.cin_total + ci) * Cout + co, acc[q][n][i]);
      }
    }
  }
}

template <int COW, int NPROD, bool STK = false, int CIW = 8>
int launch_split_wgrad_np(const SplitWgArgs& a0, hipStream_t st) {
  using C = WgCfg<COW, CIW>;
  SplitWgArgs a = a0;
  const int gy = a.ncc * a.nco;
  const int gx = split_wgrad_grid_x(a.ntiles, gy);
  const size_t smem = (size_t)C::NBUF * C::BUFB;
  auto kern = conv3d_split_wgrad_kernel<COW, NPROD, STK, CIW>;
  static SynOncePerDevice attr_done;
  if (auto once_ = attr_done.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  DetRun det;
  if (const int rc_ = syn_det_prepare(&det, &a.dw, &a.dbias, (int64_t)27 * a.cin_total * a.Cout, a.Cout, gx, st)) return rc_;
  a.det_stride = det.stride;
  hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(512), smem, st, a);
  if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
  return syn_det_finish(&det, st);
}

template <int COW>
int launch_split_wgrad(const SplitWgArgs& a, hipStream_t st) {
  if constexpr (COW == 24) {
    if (t_nprod == 6 && a.ciw == 24) return launch_split_wgrad_np<24, 6, true, 24>(a, st);
    if (t_nprod == 6) return launch_split_wgrad_np<24, 6, true>(a, st);
  }
  if constexpr (COW == 48) {
    if (t_nprod == 6 && a.ciw == 16) return launch_split_wgrad_np<48, 6, false, 16>(a, st);
  }
  return t_nprod == 9 ? launch_split_wgrad_np<COW, 9>(a, st) : launch_split_wgrad_np<COW, 6>(a, st);
}

// ---- weight gradient of the up-sampled channel range of a FOLDED decoder conv (round 6; unet.py; ext/neuron/models.py:426-444
// UpSampling3D -> concatenate -> Conv3D) in split arithmetic:
//   dwc[p][slot][ci][co] = sum over low-res voxels v of lo[v + slot - 1][ci] * dz[2 v + p][co],  slot in p + {0, 1}^3 per axis
// (8 parities x 8 of the 27 slots; synthsr_conv3d_up_unpack folds them onto the 27 original taps).  Round 3 built this as one
// workgroup per parity and lost (1.05 vs 0.72 ms at 80^3): a converted x halo then feeds 8 taps instead of 27.  Here the EIGHT
// WAVES of a workgroup each own ONE PARITY over the SAME staged x halo (16 input channels, the plain kernel's 62 KB image), so a
// converted halo feeds 64 (parity, tap) row tiles.  dz -- eight times the voxels of x -- goes through LDS in STAGES of one K step:
// two low-res x-rows (y = 2 kj, 2 kj + 1) of a z plane of the tile for all eight parities = [piece 3][parity 8][32 voxels][24
// channels] = 36 KB, double buffered; the hi-res rows are read as whole 3 KB lines.  A wave and stage: 8 row tiles (tap x 16
// channels) x the five stacked column tiles of the 24-column kernel = 80 MFMAs; wave w also stages hi-res x-row w of the NEXT
// stage (8 rows = 2 z parities x 2 low-res rows x 2 y parities; three 16-byte pieces per lane, requested one stage earlier):
// each piece is split and stored into the other buffer between the MFMAs of a row tile, in the shadow of the wave's own matrix
// instructions (a wave that converts next to ANOTHER wave's MFMA stream gets one vector instruction per ~27 cycles: measured,
// profiles/r06_upwgrad_phase_timing.txt); one barrier per stage.  The x image is single-buffered: its next tile is
// requested in the last stage of a tile and stored between two barriers at the tile boundary.  No cross-wave reduction at the
// end -- every wave flushes its own parity.  Cout = 24 nco: the column chunk is blockIdx.y / ncc.
struct SplitUpWgArgs {
  const float* lo;    // x [D0][D1][D2][Cin] (the low-resolution tensor)
  const float* dout;  // dz [2 D0][2 D1][2 D2][Cout]
  float* dwc;         // [8][27][Cin][Cout], accumulated with atomics
  int D0, D1, D2, Cin, Cout, ncc, nco, tiles1, tiles2, ntiles;
  int64_t det_stride;
};

constexpr int UW_XB = 32, UW_XPLANE = HVOX * UW_XB;            // 16 channels x 2 bytes; one piece of the x halo image
constexpr int UW_DROWB = 48, UW_PARB = 2 * TX * UW_DROWB;      // a voxel's 24 channels; one parity of a stage (2 x-rows)
constexpr int UW_DPLANE = 8 * UW_PARB, UW_DBUF = 3 * UW_DPLANE;  // one piece of a stage image; a stage image
constexpr int UW_DOFF = 3 * UW_XPLANE, UW_ZOFF = UW_DOFF + 2 * UW_DBUF, UW_LDS = UW_ZOFF + 64;
constexpr int UW_NST = 2 * TZ;                                 // stages of a tile: (z plane, x-row pair)
static_assert(UW_LDS <= 160 * 1024, "x halo image + two dz stage images fit the LDS");

__global__ __launch_bounds__(512, 1) void conv3d_split_upwgrad_kernel(const SplitUpWgArgs a) {
  constexpr int XQ = 4, DQ = 6, RT = 8, NDL = 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int par = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave's parity
  const int pz = par >> 2, py = (par >> 1) & 1, px = par & 1;
  const int g = lane >> 4, li = lane & 15, lrow = li >> 2, lq = li & 3;
  const int cc = blockIdx.y % a.ncc, oc = blockIdx.y / a.ncc;
  const TileWalk walk = tile_walk(a.ntiles);
  const int tiles0 = a.ntiles / (a.tiles1 * a.tiles2);
  const int D0 = a.D0, D1 = a.D1, D2 = a.D2, Cin = a.Cin, Cout = a.Cout;

  // operand addresses (the K <-> voxel bijection of conv3d_split_wgrad_kernel: K index 8 g + j <-> voxel (row g >> 1,
  // x = 8 (j >> 2) + 4 (g & 1) + (j & 3)) of the two x-rows of a K step); row tile q = tap (q >> 2, (q >> 1) & 1, q & 1) of the
  // parity's 2x2x2 window, whose halo origin is the parity itself (slot = p + tap per axis), channel quad lq
  const int vx = 4 * (g & 1) + lrow, vr = g >> 1;
  uint32_t aaddr[RT];
#pragma unroll
  for (int q = 0; q < RT; ++q) {
    const int hz = pz + (q >> 2), hy = py + ((q >> 1) & 1), hx = px + (q & 1);
    aaddr[q] = (uint32_t)((((hz * HY + hy + vr) * HX) + hx + vx) * UW_XB + lq * 8);
  }
  const uint32_t bvox = (uint32_t)(UW_DOFF + par * UW_PARB + (vr * TX + vx) * UW_DROWB);
  const uint32_t bbase = bvox + (uint32_t)(lq * 8);
  // stacked column tiles U3 / U4: channels 16-23 (byte 32 of the row) of piece lq >> 1 resp. of piece 2 | the zeroed slack
  const uint32_t u3off = bvox + (uint32_t)((lq >> 1) * UW_DPLANE + 32 + (lq & 1) * 8);
  const uint32_t u4off = lq < 2 ? bvox + (uint32_t)(2 * UW_DPLANE + 32 + (lq & 1) * 8) : (uint32_t)UW_ZOFF;
  const uint32_t u4hi = lq < 2 ? (uint32_t)(8 * UW_DROWB) : 0u;
  if (tid < 2) *reinterpret_cast<uint64_t*>(lds + UW_ZOFF + tid * 8) = 0ull;

  // staging of the x halo: thread = (channel quad tid & 3, in-plane halo voxel tid >> 2 of the 6 x 18), piece i = halo plane i
  const int xq = tid & 3, xr = tid >> 2;
  const int xps = D1 * D2 * Cin * 4;
  const int xhy = xr / HX, xhx = xr - xhy * HX;
  const int xrel = (xhy * D2 + xhx) * Cin * 4 + xq * 16;
  const uint32_t xmask = xr < HY * HX ? ((1u << (6 + xhy)) | (1u << (12 + xhx))) : 0xFFFFFFFFu;
  // staging of a dz stage: wave w = hi-res x-row (z parity w >> 2, low-res row (w >> 1) & 1, y parity w & 1) of the stage; piece k =
  // 64-lane third k of the row's 192 16-byte pieces: hi-res x m / 6, channel quad m % 6 (m = 64 k + lane)
  const int D1h = 2 * D1, D2h = 2 * D2;
  const int dpz = par >> 2, dyl = (par >> 1) & 1, dpy = par & 1;
  int drel[3], dxlo[3];
  uint32_t dlds[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int m = k * 64 + lane, hx = m / DQ, dq = m - hx * DQ;
    drel[k] = hx * Cout * 4 + (oc * 24 + dq * 4) * 4;
    dxlo[k] = hx >> 1;
    dlds[k] = (uint32_t)(UW_DOFF + (dpz * 4 + dpy * 2 + (hx & 1)) * UW_PARB + (dyl * TX + (hx >> 1)) * UW_DROWB + dq * 8);
  }
  const int dyp = D2h * Cout * 4;              // bytes from a hi-res row to the next
  const int dwave = (dpz * D1h + 2 * dyl + dpy) * dyp;
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.lo), 0, (int)((int64_t)D0 * D1 * D2 * Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.dout), 0, (int)((int64_t)8 * D0 * D1 * D2 * Cout * 4), 0x00020000);
  f32x4 xst[HZ], dst[2][NDL];  // dz: two register sets (stage parity)
  // (tile origins are decoded ONCE per tile -- the divisions of tile_decode were a third of the kernel's scalar instructions
  //  when every stage's request decoded its tile again)
  auto load_x = [&](int z0, int y0, int x0) {
    uint32_t bad = 0x80000000u;
    bad |= syn_oob_bits(z0 - 1, HZ, D0) | (syn_oob_bits(y0 - 1, HY, D1) << 6) | (syn_oob_bits(x0 - 1, HX, D2) << 12);
    const int base = ((((z0 - 1) * D1 + (y0 - 1)) * D2 + (x0 - 1)) * Cin + cc * 16) * 4;
#pragma unroll
    for (int i = 0; i < HZ; ++i) {
      const uint32_t vo = ((xmask | (1u << i)) & bad) ? OOB : (uint32_t)(xrel + i * xps + base);
      xst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (int)vo, 0, 0));
    }
  };
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  auto store_x = [&]() {
    if (xr >= HY * HX) return;
#pragma unroll
    for (int i = 0; i < HZ; ++i) {
      uint32_t p0, p1, p2, q0, q1, q2;
      syn_split3(xst[i][0], xst[i][1], p0, p1, p2);
      syn_split3(xst[i][2], xst[i][3], q0, q1, q2);
      const int xl = ((i * (HY * HX) + xr) * XQ + xq) * 8;
      *reinterpret_cast<u32x2*>(lds + xl) = (u32x2){p0, q0};
      *reinterpret_cast<u32x2*>(lds + UW_XPLANE + xl) = (u32x2){p1, q1};
      *reinterpret_cast<u32x2*>(lds + 2 * UW_XPLANE + xl) = (u32x2){p2, q2};
    }
  };
  auto load_dz = [&](int z0, int y0, int x0, int st, int set) {  // stage st = (z plane st >> 1, x-row pair st & 1) of the tile at (z0, y0, x0)
    const int zp = st >> 1, yy = y0 + 2 * (st & 1) + dyl;
    const bool okr = z0 + zp < D0 && yy < D1;
    const int base = ((2 * (z0 + zp) * D1h + 2 * (y0 + 2 * (st & 1))) * D2h + 2 * x0) * Cout * 4 + dwave;
#pragma unroll
    for (int k = 0; k < NDL; ++k) {
      const int vo = (okr && x0 + dxlo[k] < D2) ? base + drel[k] : (int)OOB;
      dst[set][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdo, vo, 0, 0));
    }
  };
  auto store_dz1 = [&](int set, int k, int buf) {  // piece k of register set `set` into stage image `buf`
    uint32_t p0, p1, p2, q0, q1, q2;
    syn_split3(dst[set][k][0], dst[set][k][1], p0, p1, p2);
    syn_split3(dst[set][k][2], dst[set][k][3], q0, q1, q2);
    unsigned char* d = lds + dlds[k] + buf * UW_DBUF;
    *reinterpret_cast<u32x2*>(d) = (u32x2){p0, q0};
    *reinterpret_cast<u32x2*>(d + UW_DPLANE) = (u32x2){p1, q1};
    *reinterpret_cast<u32x2*>(d + 2 * UW_DPLANE) = (u32x2){p2, q2};
  };

  f32x4 acc[RT][2];
#pragma unroll
  for (int q = 0; q < RT; ++q) acc[q][0] = acc[q][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int n2 = 0, n3 = 0;
  (void)n2;
  (void)n3;
  T2(0);
  int cz = 0, cy = 0, cx = 0, nz = 0, ny = 0, nx = 0;  // origin of the current / the next tile
  if (walk.pos < walk.end) {
    tile_decode(walk.pos, tiles0, a.tiles1, a.tiles2, cz, cy, cx);
    load_x(cz, cy, cx);
    load_dz(cz, cy, cx, 0, 0);
    load_dz(cz, cy, cx, 1, 1);
    store_x();
#pragma unroll
    for (int k = 0; k < NDL; ++k) store_dz1(0, k, 0);
  }
  __syncthreads();
  T2(1);
  for (int t = walk.pos; t < walk.end; t += walk.stride) {
    const bool more = t + walk.stride < walk.end;
    if (more) tile_decode(t + walk.stride, tiles0, a.tiles1, a.tiles2, nz, ny, nx);
    sfor<0, UW_NST>([&](auto ST) {
      constexpr int st = decltype(ST)::value, zp = st >> 1, kj = st & 1;
      T2(2);
      T3(0);
      // requests: the stage after the next one into the register set this stage's image came from; the next tile's x halo
      if constexpr (st + 2 < UW_NST) {
        load_dz(cz, cy, cx, st + 2, kj);
      } else {
        if (more) load_dz(nz, ny, nx, st + 2 - UW_NST, kj);
      }
      if constexpr (st == UW_NST - 1) {
        if (more) load_x(nz, ny, nx);
      }
      const bool next = st + 1 < UW_NST || more;  // is there a next stage to convert (its pieces: register set kj ^ 1)
      {
        constexpr uint32_t ak = (uint32_t)((zp * HY * HX + kj * 2 * HX) * UW_XB), bk = (uint32_t)(kj * UW_DBUF);
        u32x4 bfr[3][2], afr[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
          bfr[p][0] = tr_read8(lds + bbase + bk + p * UW_DPLANE, lds + bbase + bk + p * UW_DPLANE + 8 * UW_DROWB);
        bfr[0][1] = tr_read8(lds + u3off + bk, lds + u3off + bk + 8 * UW_DROWB);                   // U3
        bfr[1][1] = tr_read8(lds + u4off + (lq < 2 ? bk : 0u), lds + u4off + (lq < 2 ? bk : 0u) + u4hi);  // U4
        auto aload = [&](int q, int slot) {
#pragma unroll
          for (int p = 0; p < 3; ++p)
            afr[slot][p] = tr_read8(lds + aaddr[q] + ak + p * UW_XPLANE, lds + aaddr[q] + ak + p * UW_XPLANE + 8 * UW_XB);
        };
        aload(0, 0);
        sfor<0, RT>([&](auto Q) {
          constexpr int q = decltype(Q)::value;
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (q + 1 < RT) aload(q + 1, (q + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
          // (x piece, column tile, accumulator) in the order of the stacked 24-column kernel: smallest terms first
          auto mm = [&](int xa, int pb, int nb, int ac) {
            acc[q][ac] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afr[q & 1][xa]),
                                                                  __builtin_bit_cast(bf16x8, bfr[pb][nb]), acc[q][ac], 0, 0, 0);
          };
          mm(0, 2, 0, 0);  // x0 U2
          mm(0, 1, 1, 1);  // x0 U4
          mm(2, 0, 0, 0);  // x2 U0
          mm(2, 0, 1, 1);  // x2 U3
          mm(1, 1, 0, 0);  // x1 U1
          mm(1, 0, 1, 1);  // x1 U3
          mm(0, 1, 0, 0);  // x0 U1
          mm(0, 0, 1, 1);  // x0 U3
          mm(1, 0, 0, 0);  // x1 U0
          mm(0, 0, 0, 0);  // x0 U0
          // the next stage's pieces, one per row tile 2, 4, 6 (requested a stage ago), in the shadow of these MFMAs
          if constexpr (q >= 2 && (q & 1) == 0) {
            if (next) store_dz1(kj ^ 1, q / 2 - 1, kj ^ 1);
          }
        });
        __builtin_amdgcn_sched_barrier(0);
      }
      T2(3);
      T3(1);
      T2(4);
#ifdef SYN_SPLIT_TIMING
      ++n3;
#endif
      if constexpr (st == UW_NST - 1) {
        __syncthreads();  // everyone is done with the tile's x image
        if (more) store_x();
      }
      __syncthreads();  // the next stage's image is complete, this stage's is free
      T2(5);
    });
    cz = nz; cy = ny; cx = nx;
  }
  // ---- flush: this wave's parity; columns 16-23 = columns 0-7 + 8-15 of the stacked tile
#pragma unroll
  for (int q = 0; q < RT; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[q][1][i] += __shfl_down(acc[q][1][i], 8, 16);
  float* dwp = a.dwc + (size_t)blockIdx.x * a.det_stride + (size_t)par * 27 * Cin * Cout;
#pragma unroll
  for (int q = 0; q < RT; ++q) {
    const int slot = ((pz + (q >> 2)) * 3 + py + ((q >> 1) & 1)) * 3 + px + (q & 1);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int col = n * 16 + li;
      if (col >= 24) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ci = cc * 16 + 4 * g + i;
        atomicAdd(dwp + ((int64_t)slot * Cin + ci) * Cout + oc * 24 + col, acc[q][n][i]);
      }
    }
  }
}

}  // namespace

// weight gradient of the up-sampled channel range of a folded decoder conv (six products; SYNTHSR_EINVAL = shape not covered, the
// caller takes the fp32-MFMA path); dwc [8][27][Cin][Cout] accumulated (the caller zeroes it)
extern "C" __attribute__((visibility("hidden"))) int syn_split_upwgrad(const float* lo, const float* dout, float* dwc,
                                                                        const int s[3], int Cin, int Cout, int nprod,
                                                                        hipStream_t st) {
  if (nprod != 6 || (Cin % 16) != 0 || (Cout % 24) != 0) return SYNTHSR_EINVAL;
  const int64_t vox = (int64_t)s[0] * s[1] * s[2];
  if (vox * Cin * 4 >= (1ll << 31) || 8 * vox * Cout * 4 >= (1ll << 31)) return SYNTHSR_EINVAL;
  SplitUpWgArgs a;
  a.lo = lo;
  a.dout = dout;
  a.dwc = dwc;
  a.D0 = s[0]; a.D1 = s[1]; a.D2 = s[2];
  a.Cin = Cin; a.Cout = Cout;
  a.ncc = Cin / 16;
  a.nco = Cout / 24;
  a.tiles1 = (s[1] + TY - 1) / TY;
  a.tiles2 = (s[2] + TX - 1) / TX;
  a.ntiles = ((s[0] + TZ - 1) / TZ) * a.tiles1 * a.tiles2;
  const int gy = a.ncc * a.nco;
  const int gx = split_wgrad_grid_x(a.ntiles, gy);
  static SynOncePerDevice attr_done;
  if (auto once_ = attr_done.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_split_upwgrad_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)UW_LDS);
  }
  DetRun det;
  float* no_dbias = nullptr;
  if (const int rc_ = syn_det_prepare(&det, &a.dwc, &no_dbias, (int64_t)8 * 27 * Cin * Cout, Cout, gx, st)) return rc_;
  a.det_stride = det.stride;
  hipLaunchKernelGGL(conv3d_split_upwgrad_kernel, dim3(gx, gy), dim3(512), (size_t)UW_LDS, st, a);
  if (hipGetLastError() != hipSuccess) return SYNTHSR_ELAUNCH;
  return syn_det_finish(&det, st);
}

// include/synthsr_hip_tuning.h: host restatement (the same function the kernels call) of the tile schedule, for tests
extern "C" int synthsr_split_tile_schedule(int kernel, int ntiles, int ny, int block_x, int block_yz, int out[4]) {
  if (ntiles < 1 || ny < 1 || block_x < 0 || block_yz < 0 || !out) return SYNTHSR_EINVAL;
  int gx;
  if (kernel == 0) gx = split_grid_x(ntiles, ny);             // forward / data gradient: ny = output-channel chunks
  else if (kernel == 1) gx = split_wgrad_grid_x(ntiles, ny);  // weight gradient: ny = input-channel chunks x column groups
  else if (kernel == 2) gx = split_upfwd_grid_x(ntiles, ny);  // folded forward: ny = parity groups (2 or 4)
  else return SYNTHSR_EINVAL;
  out[0] = gx;
  if (block_x >= gx) return SYNTHSR_EINVAL;
  const TileWalk w = tile_walk_of(gx, block_x, gx * block_yz, ntiles);
  out[1] = w.pos;
  out[2] = w.end;
  out[3] = w.stride;
  return SYNTHSR_OK;
}

// called by conv3d.hip's dispatcher when the plan of the layer says `split` (weights packed in the split layout by pack_value).
// stats != null: BatchNorm batch statistics (mean | biased variance) of the output, from per-workgroup sums in `partial`
// (room for 512 x 2 Cout floats)
// upm = 2: the data gradient of a folded decoder conv (`in` = dz on the 2x grid, s = the low-resolution grid, wp = 8 parity sets)
// does the six-product forward / data-gradient launch of this plain layer run as split-K halves (reported by synthsr_conv3d_plan)?
extern "C" __attribute__((visibility("hidden"))) int syn_split_fwd_halves(const int s[3], int Cin, int nchunks, int stacked,
                                                                           int nprod) {
  const int ntiles = ((s[0] + TZ - 1) / TZ) * ((s[1] + TY - 1) / TY) * ((s[2] + TX - 1) / TX);
  if (stacked || nprod != 6 || split_uses_fwd3(ntiles, nchunks, 3, 0)) return 0;
  return split_fwd2_uses_halves(split_grid_x(ntiles, nchunks), nchunks, Cin / 8) ? 1 : 0;
}

extern "C" __attribute__((visibility("hidden"))) int syn_split_fwd(const float* in, const float* wp, const float* bias,
                                                                    const float* addend, float* out, const int s[3], int Cin,
                                                                    int Cout, int mt, int nchunks, int act, float* stats,
                                                                    float* partial, int upm, int stacked, int nprod,
                                                                    hipStream_t st) {
  if (nprod != 6 && nprod != 9) return SYNTHSR_EINVAL;
  t_nprod = nprod;
  if ((Cin % 8) != 0 || (Cout % 4) != 0 || mt < 1 || mt > 3 || nchunks < 1 || (upm != 0 && upm != 2)) return SYNTHSR_EINVAL;
  if (stacked && (Cout != 24 || mt != 2 || nchunks != 1 || upm != 0 || t_nprod != 6)) return SYNTHSR_EINVAL;
  if (stats && (!partial || addend || act == 2 || upm)) return SYNTHSR_EINVAL;
  if (upm && (bias || addend || act != 0)) return SYNTHSR_EINVAL;
  const int64_t vox = (int64_t)s[0] * s[1] * s[2];
  if ((upm ? 8 : 1) * vox * Cin * 4 >= (1ll << 31) || vox * Cout * 4 >= (1ll << 31)) return SYNTHSR_EINVAL;
  SplitFwdArgs a;
  a.in = in;
  a.wp = reinterpret_cast<const u32x4*>(wp);
  a.bias = bias;
  a.addend = addend;
  a.out = out;
  a.stats_partial = stats ? partial : nullptr;
  a.D0 = s[0]; a.D1 = s[1]; a.D2 = s[2];
  a.Cin = Cin; a.Cout = Cout; a.ncc = Cin / 8;
  a.tiles1 = (s[1] + TY - 1) / TY;
  a.tiles2 = (s[2] + TX - 1) / TX;
  a.ntiles = ((s[0] + TZ - 1) / TZ) * a.tiles1 * a.tiles2;
  a.act = act;
  a.stacked = stacked;
  const int gx = split_grid_x(a.ntiles, nchunks);
  int rc;
  if (stats) {
    rc = mt == 1 ? launch_split_fwd<1, true>(a, gx, nchunks, st)
                 : (mt == 2 ? launch_split_fwd<2, true>(a, gx, nchunks, st) : launch_split_fwd<3, true>(a, gx, nchunks, st));
    if (rc != SYNTHSR_OK) return rc;
    const int gcols = (!stacked && t_nprod == 6 && split_uses_fwd3(a.ntiles, nchunks, mt, Cout)) ? split_upfwd_grid_x(a.ntiles, nchunks) : gx;  // workgroup columns that wrote partials
    return synthsr_bn_stats_from_partials(partial, gcols, vox, Cout, stats, (synthsr_stream_t)st);
  }
  if (upm == 2)
    return mt == 1 ? launch_split_fwd<1, false, 2>(a, gx, nchunks, st)
                   : (mt == 2 ? launch_split_fwd<2, false, 2>(a, gx, nchunks, st) : launch_split_fwd<3, false, 2>(a, gx, nchunks, st));
  rc = mt == 1 ? launch_split_fwd<1, false>(a, gx, nchunks, st)
               : (mt == 2 ? launch_split_fwd<2, false>(a, gx, nchunks, st) : launch_split_fwd<3, false>(a, gx, nchunks, st));
  return rc;
}

// forward pass of the up-sampled channel range of a folded decoder conv: lo [s][Cin] -> out [2 s][Cout] (Cout <= 48: one co-chunk),
// wp = 8 parity sets in the split layout; act 0 / 1, optional bias and addend (indexed like the output)
extern "C" __attribute__((visibility("hidden"))) int syn_split_upfwd(const float* lo, const float* wp, const float* bias,
                                                                      const float* addend, float* out, const int s[3], int Cin,
                                                                      int Cout, int mt, int act, int nprod, hipStream_t st) {
  if (nprod != 6 && nprod != 9) return SYNTHSR_EINVAL;
  t_nprod = nprod;
  if ((Cin % 8) != 0 || (Cout % 4) != 0 || mt < 1 || mt > 3 || Cout > 16 * mt || (act != 0 && act != 1)) return SYNTHSR_EINVAL;
  const int64_t vox = (int64_t)s[0] * s[1] * s[2];
  if (vox * Cin * 4 >= (1ll << 31) || 8 * vox * Cout * 4 >= (1ll << 31)) return SYNTHSR_EINVAL;
  SplitFwdArgs a;
  a.in = lo;
  a.wp = reinterpret_cast<const u32x4*>(wp);
  a.bias = bias;
  a.addend = addend;
  a.out = out;
  a.stats_partial = nullptr;
  a.D0 = s[0]; a.D1 = s[1]; a.D2 = s[2];
  a.Cin = Cin; a.Cout = Cout; a.ncc = Cin / 8;
  a.tiles1 = (s[1] + TY - 1) / TY;
  a.tiles2 = (s[2] + TX - 1) / TX;
  a.ntiles = ((s[0] + TZ - 1) / TZ) * a.tiles1 * a.tiles2;
  a.act = act;
  a.stacked = 0;
  if (mt == 1) return launch_split_upfwd<1, 4>(a, st);
  if (mt == 2) return launch_split_upfwd<2, 4>(a, st);
  return launch_split_upfwd<3, 2>(a, st);
}

// weight gradient of the input-channel range [ci_off, ci_off + Cin) of a layer with cin_total input channels (+ optionally the
// bias gradient); SYNTHSR_EINVAL = shape not covered, the caller takes the fp32-MFMA path
extern "C" __attribute__((visibility("hidden"))) int syn_split_wgrad(const float* in, const float* dout, float* dw, float* dbias,
                                                                      const int s[3], int cin_total, int ci_off, int Cin,
                                                                      int Cout, int nprod, hipStream_t st) {
  if (nprod != 6 && nprod != 9) return SYNTHSR_EINVAL;
  t_nprod = nprod;
  if ((Cin % 8) != 0 || (Cout % 24) != 0) return SYNTHSR_EINVAL;
  const int64_t vox = (int64_t)s[0] * s[1] * s[2];
  if (vox * Cin * 4 >= (1ll << 31) || vox * Cout * 4 >= (1ll << 31)) return SYNTHSR_EINVAL;
  SplitWgArgs a;
  a.in = in;
  a.dout = dout;
  a.dw = dw;
  a.dbias = dbias;
  a.D0 = s[0]; a.D1 = s[1]; a.D2 = s[2];
  a.Cin = Cin; a.Cout = Cout; a.cin_total = cin_total; a.ci_off = ci_off;
  // 48-wide workgroups where they divide Cout (measured in round 3: 10 % faster than 2 x 24 padded column chunks)
  const bool c48 = (Cout % 48) == 0;
  // ... with 16 input channels each where those divide Cin (six products; option 12 bit 2 switches it off for A/B runs)
  a.ciw = (c48 && (Cin % 16) == 0 && t_nprod == 6) ? 16 : 8;
  // the stacked 24-column kernel with all 24 input channels in one workgroup (option 12 bit 3 switches it off for A/B runs)
  if (!c48 && Cin == 24 && t_nprod == 6) a.ciw = 24;
  a.ncc = Cin / a.ciw;
  a.nco = c48 ? Cout / 48 : Cout / 24;
  a.tiles1 = (s[1] + TY - 1) / TY;
  a.tiles2 = (s[2] + TX - 1) / TX;
  a.ntiles = ((s[0] + TZ - 1) / TZ) * a.tiles1 * a.tiles2;
  a.det_stride = 0;
  return c48 ? launch_split_wgrad<48>(a, st) : launch_split_wgrad<24>(a, st);
}
