// Implicit-GEMM convolution on CDNA4 bf16 MFMA (v_mfma_f32_32x32x16_bf16), NHWC activations.
//
// Replaces the nn.Conv2d call sites of the AdaptSegNet trunk/head (adaptseg.py:85,89,93,97,119-121).
//   GEMM view:  D[m][n] = sum_k A[m][k] * Wt[n][k]
//     m = output pixel (n_img, oh, ow)            M = B*Ho*Wo
//     n = output channel                           N = Cout
//     k = (ci chunk, tap, ci % 64)                 K = ntaps*Cin, one 64-wide K step = one tap of one 64-channel chunk
//   A is gathered on the fly (im2col never materialised): row m, tap t reads the 128-byte channel
//   slab of input pixel (oh*stride+dy[t], ow*stride+dx[t]); out-of-image taps and rows beyond the sample read
//   a 16-byte zero page instead (a select on the source address, so the gather stays branch-free).
//
// Block = 4 or 8 waves (WM x WN), tile BM x BN x 64.  The K loop is an S-deep LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4, no register staging) with counted vmcnt and one raw s_barrier per K step.
// LDS rows are 128 B (64 bf16); the 16-byte chunk index is XOR-swizzled with (row>>1)&7 -- on the SOURCE
// address, because the DMA writes a lane-linear image -- so that the ds_read_b128 fragment reads of 32
// consecutive rows are bank-conflict free.
//
// M tiles never straddle samples (tile -> (sample, tile in sample)), so a sample's rows are computed the same
// way whatever batch it rides in.
//
// Epilogue: (1) DAM statistics -- per-channel sum / sum of squares of (x - shift[c]) taken from the fp32
// accumulators, one "unit" per 32-row MFMA block (columns live on lanes, rows in registers: 16 adds + one
// cross-half shuffle), written to a slab that diga_bn_finalize sums in fixed order (deterministic, no atomics,
// independent of the tile shape);  (2) the raw output is staged through LDS and stored as 16-byte NHWC rows.
//
// Verification mode (desc.nplanes = 2 / 3): the SAME kernel with every operand given as 2 / 3 bf16 planes whose sum
// is the fp32 value (hi = bf16(x), lo = bf16(x - hi), ...).  The K loop then runs over the operand-plane products
// (a, b) with a + b < nplanes -- 3 terms ("bf16x3", ~2^-17 per product) or 6 terms ("bf16x6", fp32-grade) -- as extra
// K ranges, small terms first, into the same fp32 accumulators; output and statistics are taken in fp32.  This is the
// fp32-exact path the end-to-end parity gate runs (the reference is fp32 end to end, ttda:1634-1645).
#include "common.h"
#include "dam_affine.h"
#include <mutex>
#include <type_traits>

namespace {

struct ConvKParams {
  const void* in;
  const void* wgt;
  void* out;
  float* slab;
  const float* shift;
  const float* ep_scale;     // EPI_FUSED_APPLY: affine [C] or [nsamples][C] (ep_affine_stride = C), residual bf16 [M][Cout]
  const float* ep_shift;
  const void* ep_res;
  int ep_affine_stride, ep_relu;
  int H, W, Cin, Ho, Wo, Cout, stride, ntaps;
  int M, K, cout_pad, hw, tiles_per_sample, units_per_sample, nk_total, nk_per_split, mtiles, ntiles_n;
  // operand planes (verification mode): nk_total = nterms * nk_plane K steps; term t multiplies plane term_ab[t] & 0xff of
  // the input (in + plane * in_plane_stride elements) with plane term_ab[t] >> 8 of the weights.  nterms = 1: plain bf16.
  int nterms, nk_plane, in_plane_stride, wgt_plane_stride;
  int term_ab[6];
  // fused DAM finalize (diga_conv_igemm_bf16_dam): the LAST tile to finish of each (sample, N tile) group reduces that
  // sample's slab columns [n0, n0 + BN) exactly as diga_bn_finalize(per_sample) does and writes scale / shift.
  // fused DAM apply on the INPUT operand (diga_conv_igemm_bf16_fused_in, AMODE > 0): `in` is the producer's RAW bf16 output;
  // A = bf16(relu(fma(raw, a_scale[c], a_shift[c]) [+ a_res])) is formed in registers on its way to LDS (the arithmetic of
  // bn_apply_kernel<0> / <1>), so the normalised activation is never written and read back.  AMODE 2 (1x1 convs) also
  // writes A to a_out -- the block input the NEXT residual needs -- from the N-tile-0 blocks.
  const float* a_scale;      // [Cin] or [samples][Cin] (a_affine_stride = Cin)
  const float* a_shift;
  const void* a_res;         // bf16 [rows][Cin], AMODE 2
  void* a_out;               // bf16 [rows][Cin], AMODE 2
  int a_affine_stride;
  unsigned* fin_ticket;      // NULL = off; [samples * ntiles_n] counters, zero before the launch, left zero by it
  const float* fin_rvar;     // running_var, weight, bias [Cout] (running_mean = shift)
  const float* fin_weight;
  const float* fin_bias;
  float* fin_scale;          // [samples][Cout]
  float* fin_shift;
  float fin_lam, fin_eps;
  double fin_n;              // elements per sample and channel
  int dydx[DIGA_MAX_TAPS];   // (dy << 16) | (dx & 0xffff): 32-bit so that the uniform per-tap lookup is a scalar load
};

// 16 zero bytes: the source of every out-of-image / out-of-range 16-byte chunk of the gather
__device__ __attribute__((aligned(16))) unsigned int g_zero16[4];

// Per-tile phase clock (profiles/experiments/tile_clock): compiled in with -DDIGA_TILE_CLOCK only, never in the product build.
#ifdef DIGA_TILE_CLOCK
__device__ unsigned long long g_tile_clock[16 * 16384];
#define DIGA_TCLK(i) tclk[i] = wall_clock64()   /* kept in registers: a store inside the K loop would disturb its vmcnt counting */
#else
#define DIGA_TCLK(i) \
  do {               \
  } while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// ---- DAM statistics of one unit (32 consecutive rows of a sample x one channel), summed in ONE fixed tree whatever the
// tile shape, the MFMA shape or the kernel, so that a unit's bits never depend on who produced it:
//   4-row group g (rows 4g .. 4g+3), d = x - shift[c]:   S_g = (d0 + d2) + (d1 + d3),   Q_g = fma(d2, d2, d0*d0) + fma(d3, d3, d1*d1)
//   unit = [(g0 + g4) + (g2 + g6)] + [(g1 + g5) + (g3 + g7)]      (S and Q alike)
// Both accumulator layouts keep whole 4-row groups in one lane (32x32x16: registers 4q .. 4q+3 of lane half hh = group 2q + hh;
// 16x16x32: lane group fq of the unit's two 16-row tiles = groups fq and 4 + fq), so a group is packed fp32 arithmetic on register
// pairs (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) and the rest of the tree is one or two cross-lane adds.
// (S_g, Q_g) of one group; valid = bit i set when row i of the group is inside the sample (rows beyond it contribute d = 0)
// (FULL: the whole unit is inside the sample -- a wave-uniform fact, so callers branch on it ONCE per unit instead of
// letting the compiler turn the masking into selects on every element)
template <bool FULL>
__device__ __forceinline__ f32x2 stat_group(float x0, float x1, float x2, float x3, float k, unsigned valid) {
  const f32x2 kk = {k, k};
  f32x2 A = f32x2{x0, x1} - kk, B = f32x2{x2, x3} - kk;
  if constexpr (!FULL) {
    A[0] = (valid & 1u) ? A[0] : 0.f;
    A[1] = (valid & 2u) ? A[1] : 0.f;
    B[0] = (valid & 4u) ? B[0] : 0.f;
    B[1] = (valid & 8u) ? B[1] : 0.f;
  }
  const f32x2 P = A + B, F = __builtin_elementwise_fma(B, B, A * A);     // (scalar adds / FMAs instead: same time, same-box A/B)
  // (the two closing adds get SLP-packed into one v_pk_add_f32 fed by three v_mov_b32; written as opaque scalar v_add_f32 the 48
  // moves per unit disappear and NOTHING changes: 108-110 vs 106-107 us on layer3's conv3, round 3)
  return f32x2{P[0] + P[1], F[0] + F[1]};
}
// Cross-lane part.  gfx950 lane swaps instead of __shfl_xor's ds_bpermute (an LDS round trip per call); inline asm because the
// __builtin_amdgcn_permlane*_swap builtins of this toolchain return the first result twice.
// t = (S, Q) partial of this lane -> lanes 0-31: S(lane) + S(lane + 32), lanes 32-63: Q(lane - 32) + Q(lane): ONE swap folds
// both sums (v_permlane32_swap exchanges lanes 32-63 of its first operand with lanes 0-31 of its second)
__device__ __forceinline__ float stat_fold32(f32x2 t) {
  float a = t[0], b = t[1];
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // a = [S.lo | Q.lo], b = [S.hi | Q.hi]
  return a + b;
}
// x + (x of lane ^ 16) in every lane (v_permlane16_swap exchanges the odd 16-lane rows of a with the even rows of b)
__device__ __forceinline__ float xor_add16(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // a = even rows twice, b = odd rows twice
  return a + b;
}

// Epilogue modes: 0 = bf16 output + statistics (the trunk), 1 = fp32 output (head GEMM / split-K partials),
// 2 = bf16 output only, 3 = statistics only (nothing stored: pass 1 of a recomputed conv3), 4 = fused DAM apply:
// out = relu(acc*scale[c] + shift[c] + residual) in bf16, straight from the fp32 accumulators (pass 2).
// 5 = fp32 output + statistics (verification mode: nothing is rounded to bf16 between the accumulators and the next stage).
enum { EPI_BF16_STATS = 0, EPI_F32 = 1, EPI_BF16 = 2, EPI_STATS_ONLY = 3, EPI_FUSED_APPLY = 4, EPI_F32_STATS = 5 };

// MF = the MFMA shape: 16 = v_mfma_f32_16x16x32_bf16 (the build default), 32 = v_mfma_f32_32x32x16_bf16 (-DDIGA_MF=32, kept
// for A/B runs).  Same FLOPs per cycle, the same LDS bytes and -- measured, tests/test_gpu_kernels.py -- the same bits in the
// fp32 accumulators; under load the chip holds a higher clock on the 16x16x32 form (MI355X_MICROARCH.md 'DVFS give-back'
// item 7): same-box A/B at 15 frames, 256x256 tiles: l3.c2 171 -> 147 us, l4.c2 544 -> 477 us (1244 TFLOP/s).
#ifndef DIGA_MF
#define DIGA_MF 16
#endif
template <int BM, int BN, int WM, int WN, int S, int BK, bool ILV, int EPI, int AMODE = 0, int MF = DIGA_MF>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_kernel(const ConvKParams p) {
  static_assert(AMODE == 0 || (S == 2 && !ILV && EPI == EPI_BF16_STATS), "the fused-input path: 2-deep ring, burst issue, trunk epilogue");
  static_assert(MF == 32 || MF == 16, "MFMA shape");
  constexpr bool OUT_F32 = EPI == EPI_F32 || EPI == EPI_F32_STATS;
  constexpr bool STATS = EPI == EPI_BF16_STATS || EPI == EPI_STATS_ONLY || EPI == EPI_F32_STATS;
  constexpr int NW = WM * WN, NTHR = 64 * NW;        // 4 or 8 waves
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / MF, NT = WTN / MF;      // MFMA tiles per wave
  constexpr int AR = MF == 32 ? 16 : 4;            // accumulator registers per MFMA tile
  using acc_t = std::conditional_t<MF == 32, f32x16, f32x4>;
  // K step = BK channels (64 or 32): LDS rows of ROWB bytes = CPRW 16-byte chunks; one 1-KiB LDS-DMA wave-instruction
  // covers RPI rows.  BK = 32 halves the stage, so twice as many stages (loads in flight) fit the same LDS.
  constexpr int ROWB = BK * 2, CPRW = BK / 8, RPI = 64 / CPRW;
  constexpr int A_CH = BM / (RPI * NW), B_CH = BN / (RPI * NW);   // LDS-DMA wave-instructions per wave per K step
  constexpr int LPT = A_CH + B_CH;                   // loads per tile per wave (the vmcnt unit)
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
  // S = LDS ring depth; S-1 K steps of loads stay in flight
  constexpr int D = S - 1;
  static_assert((NW == 4 || NW == 8) && (BK == 64 || BK == 32) && MT >= 1 && NT >= 1 && A_CH >= 1 && B_CH >= 1 && S >= 2 && S <= 4, "bad tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the ONLY LDS object (cdna guide 5, trap 4a)

  const int tid = threadIdx.x, lane = tid & 63;
#ifdef DIGA_TILE_CLOCK
  unsigned long long tclk[8];
#endif
  DIGA_TCLK(0);
#ifdef DIGA_TILE_CLOCK
  if (tid == 0 && blockIdx.x < 16384) {
    g_tile_clock[(size_t)blockIdx.x * 16 + 8] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
    g_tile_clock[(size_t)blockIdx.x * 16 + 9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
  }
#endif
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // lane -> (column of the MFMA tile, row group): 32x32x16: col = lane & 31, hh = lane >> 5; 16x16x32: col = lane & 15, fq = lane >> 4
  const int l31 = lane & (MF - 1), hh = lane / MF;
  // accumulator element r of MFMA tile (a, b): its row / column inside the wave's WTM x WTN tile
  auto acc_row = [&](int a, int r) { return MF == 32 ? a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh : a * 16 + 4 * hh + r; };
  auto acc_col = [&](int b) { return b * MF + l31; };

  // XCD-aware tile order: blocks are dealt round-robin over the 8 XCDs, so give each XCD a contiguous
  // run of logical tile ids (N tiles of one M tile share the gathered A rows in that XCD's L2).
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mtile = bid / p.ntiles_n, ntile = bid - mtile * p.ntiles_n;
  // M tiles never straddle samples: tile -> (sample, tile within the sample).  A sample's rows are tiled from
  // its own row 0, so its results (and its 32-row statistics units) do not depend on the batch it rides in.
  const int n_img = mtile / p.tiles_per_sample;
  const int r0 = (mtile - n_img * p.tiles_per_sample) * BM;     // first row of the tile inside the sample
  const int m_base = n_img * p.hw;                              // global output row of the sample's row 0
  const int n0 = ntile * BN;
  const int ks_begin = blockIdx.y * p.nk_per_split;
  const int ks_end = min(ks_begin + p.nk_per_split, p.nk_total);
  const int nt = ks_end - ks_begin;

  // ---- gather state.  One LDS-DMA wave-instruction (global_load_lds_dwordx4) writes 1 KiB = RPI tile rows of
  // ROWB bytes, lane l -> row l / CPRW, 16-byte slot l % CPRW.  The LDS image is lane-linear, so the XOR swizzle is
  // applied on the SOURCE: slot s of row r holds channel chunk s ^ swz(r), swz(r) = (r>>1)&7 for 128-byte rows,
  // (r>>2)&3 for 64-byte rows: 16 consecutive rows then cover all 16 bank quads (cdna guide 5.4 rule 21).
  // Wave w owns row groups w, w+NW, ... (rows RPI*(w + NW*i) + l/CPRW); those rows are fixed for the whole K loop.
  // (64-byte rows under 16x16x32 reads: a lane group of one ds_read_b128 takes rows 0-3 and 12-15 of chunk fq and rows 4-11
  // of chunk fq + 1; XOR 3 on rows 8-15 puts its 16 lanes on the 16 bank quads)
  auto swz = [](int row) { return BK == 64 ? (row >> 1) & 7 : MF == 32 ? (row >> 2) & 3 : ((row >> 3) & 1) * 3; };
  const int lrow = RPI * wave + lane / CPRW;
  const int cch = (lane % CPRW) ^ swz(lrow);                    // rows advance by multiples of 16 with i: swz is i-invariant
  // K steps [ks_begin, ks_end) of the launch's nterms * nk_plane: term = step / nk_plane picks the operand planes
  int is_term = ks_begin / p.nk_plane;
  int is_kin = ks_begin - is_term * p.nk_plane;                 // K step inside the term
  const unsigned short* in = reinterpret_cast<const unsigned short*>(p.in) + (size_t)(p.term_ab[is_term] & 0xff) * p.in_plane_stride;
  const unsigned short* wgt = reinterpret_cast<const unsigned short*>(p.wgt) + (size_t)(p.term_ab[is_term] >> 8) * p.wgt_plane_stride;
  const unsigned short* const zero = reinterpret_cast<const unsigned short*>(g_zero16);
  int pixoff[A_CH], ih0[A_CH], iw0[A_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int rem = r0 + lrow + RPI * NW * i;
    if (rem < p.hw) {
      const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
      ih0[i] = oh * p.stride;
      iw0[i] = ow * p.stride;
      pixoff[i] = ((n_img * p.H + ih0[i]) * p.W + iw0[i]) * p.Cin + cch * 8;
    } else {
      ih0[i] = -100000;  // every tap lands outside the image -> zeros
      iw0[i] = 0;
      pixoff[i] = 0;
    }
  }
  // weight rows: a fixed 32-bit BYTE offset per lane, added to a wave-uniform base that carries the K step -- the address
  // then needs no vector arithmetic per piece (scalar base + vector offset form of the load)
  unsigned wrow_off[B_CH];
#pragma unroll
  for (int j = 0; j < B_CH; ++j) wrow_off[j] = (unsigned)(((n0 + lrow + RPI * NW * j) * p.K + cch * 8) * 2);

  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  // K steps are issued in increasing order, so (chunk, tap) advance incrementally and the tap's offset word is
  // fetched one issue ahead (a scalar load whose latency would otherwise sit between the barrier and the DMAs).
  // K order = [channel chunk][tap][64]: the ntaps steps of one 64-channel chunk run back to back, so the taps'
  // overlapping pixel lines are re-read while they are still in L2 (a 3x3 tile touches ~3x, not 9x, its rows).
  int is_chunk, is_tap;
  {
    const int q = BK == 64 ? is_kin : is_kin >> 1;              // index of the 64-wide (chunk, tap) step
    is_chunk = q / p.ntaps;
    is_tap = q - is_chunk * p.ntaps;
  }
  int is_dd = p.dydx[is_tap];
  // fused-input path (AMODE > 0): this lane's A chunks of the K step in flight, the step's channel affine, validity bits
  [[maybe_unused]] u32x4 a_raw[A_CH], a_rsd[A_CH];
  [[maybe_unused]] f32x4 a_sc[2], a_sh[2];
  [[maybe_unused]] unsigned a_mask = 0u;
  [[maybe_unused]] int a_tapoff = 0;
  [[maybe_unused]] auto a_store = [&](int stage) {
    // A = bf16(relu(fma(raw, scale, shift) [+ residual])): bn_apply_kernel's arithmetic, rounded once, exactly the value the
    // unfused path stores and re-reads; out-of-image taps and rows beyond the sample stay zero (zero padding)
    unsigned char* const dA = smem + stage * STAGE_BYTES + wave * 1024 + lane * 16;
    const float sc[8] = {a_sc[0][0], a_sc[0][1], a_sc[0][2], a_sc[0][3], a_sc[1][0], a_sc[1][1], a_sc[1][2], a_sc[1][3]};
    const float sh[8] = {a_sh[0][0], a_sh[0][1], a_sh[0][2], a_sh[0][3], a_sh[1][0], a_sh[1][1], a_sh[1][2], a_sh[1][3]};
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      float v[8], q[8];
      unpack_bf16x8(a_raw[i], v);
      if constexpr (AMODE == 2) unpack_bf16x8(a_rsd[i], q);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float y = fmaf(v[e], sc[e], sh[e]);
        if constexpr (AMODE == 2) y += q[e];
        v[e] = fmaxf(y, 0.f);
      }
      const bool ok = (a_mask >> i) & 1u;
      const u32x4 o = ok ? pack_bf16x8(v) : u32x4{0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(dA + i * NW * 1024) = o;
      if constexpr (AMODE == 2) {
        if (ok && ntile == 0) *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(p.a_out) + (pixoff[i] + a_tapoff)) = o;
      }
    }
  };

  // One K step's loads are LPT pieces (A pieces, then B pieces).  issue_pieces<lo, hi> issues pieces [lo, hi) of the next
  // K step; advance_next() moves on to the following step.  ILV kernels spread the pieces over the MFMA groups of the
  // running step (an LDS-DMA piece costs the issuing wave ~100 cycles: issued in one burst right after the barrier,
  // both waves of a SIMD burst together and the MFMA pipe idles).
  auto issue_pieces = [&](int stage, auto lo_c, auto hi_c) {
    constexpr int lo = decltype(lo_c)::value, hi = decltype(hi_c)::value;
    const int kk = is_kin * BK;
    const int c0 = is_chunk * 64 + (BK == 64 ? 0 : (is_kin & 1) * 32);
    const int dy = is_dd >> 16, dx = (int)(short)(is_dd & 0xffff);
    const int tapoff = (dy * p.W + dx) * p.Cin + c0;
    unsigned char* const dA = smem + stage * STAGE_BYTES + wave * 1024;
    unsigned char* const dB = dA + A_BYTES;
    if constexpr (AMODE > 0) {
      // register-staged A: raw (+ residual) chunks and this K step's 8 channels of scale / shift; transformed and written to
      // LDS by a_store() once they have landed
      a_tapoff = tapoff;
      a_mask = 0u;
      const float* sp = p.a_scale + (size_t)n_img * p.a_affine_stride + c0 + cch * 8;
      const float* hp = p.a_shift + (size_t)n_img * p.a_affine_stride + c0 + cch * 8;
      a_sc[0] = *reinterpret_cast<const f32x4*>(sp), a_sc[1] = *reinterpret_cast<const f32x4*>(sp + 4);
      a_sh[0] = *reinterpret_cast<const f32x4*>(hp), a_sh[1] = *reinterpret_cast<const f32x4*>(hp + 4);
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        const bool ok = (unsigned)(ih0[i] + dy) < (unsigned)p.H && (unsigned)(iw0[i] + dx) < (unsigned)p.W;
        a_mask |= ok ? 1u << i : 0u;
        const unsigned short* src = ok ? in + (pixoff[i] + tapoff) : zero;
        a_raw[i] = *reinterpret_cast<const u32x4*>(src);
        if constexpr (AMODE == 2) {
          const unsigned short* rsrc = ok ? reinterpret_cast<const unsigned short*>(p.a_res) + (pixoff[i] + tapoff) : zero;
          a_rsd[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(rsrc));   // read for the last time
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        if (i < lo || i >= hi) continue;
#ifdef DIGA_EXP_SKIPA   /* experiment only (profiles/experiments/skipa): 2 of 3 taps fetch the zero page instead of their A rows */
        const bool ok = (unsigned)(ih0[i] + dy) < (unsigned)p.H && (unsigned)(iw0[i] + dx) < (unsigned)p.W && (p.ntaps == 1 || dx == 0);
#else
        const bool ok = (unsigned)(ih0[i] + dy) < (unsigned)p.H && (unsigned)(iw0[i] + dx) < (unsigned)p.W;
#endif
        const unsigned short* src = ok ? in + (pixoff[i] + tapoff) : zero;
        __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(dA + i * NW * 1024), 16, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < B_CH; ++j) {
      if (A_CH + j < lo || A_CH + j >= hi) continue;
      __builtin_amdgcn_global_load_lds((gbl_ptr)(reinterpret_cast<const char*>(wgt + kk) + wrow_off[j]), (lds_ptr)(dB + j * NW * 1024), 16, 0, 0);
    }
  };
  auto advance_next = [&]() {
    if (++is_kin == p.nk_plane) {          // next operand-plane product (verification mode; never taken with nterms = 1)
      is_kin = 0;
      is_chunk = 0;
      is_tap = 0;
      is_dd = p.dydx[0];
      if (++is_term < p.nterms) {
        in = reinterpret_cast<const unsigned short*>(p.in) + (size_t)(p.term_ab[is_term] & 0xff) * p.in_plane_stride;
        wgt = reinterpret_cast<const unsigned short*>(p.wgt) + (size_t)(p.term_ab[is_term] >> 8) * p.wgt_plane_stride;
      }
    } else if (BK == 64 || !(is_kin & 1)) {
      if (++is_tap == p.ntaps) {
        is_tap = 0;
        ++is_chunk;
      }
      is_dd = p.dydx[is_tap];
    }
  };
  auto issue_next = [&](int stage) {
    issue_pieces(stage, std::integral_constant<int, 0>{}, std::integral_constant<int, LPT>{});
    advance_next();
  };

  acc_t acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < AR; ++r) acc[a][b][r] = 0.f;

  // fragment reads: lane (l31, hh) reads the 16-byte chunk (KCH * s + hh) of row l31 of each MFMA tile; the row's swizzle
  // depends on l31 only (tile origins are multiples of 16 rows)
  const int fr_swz = swz(l31);
  const int fa_row = (wm * WTM + l31) * ROWB, fb_row = A_BYTES + (wn * WTN + l31) * ROWB;

  // ---- K loop: S-deep LDS ring filled by LDS-DMA, one raw barrier per K step, counted vmcnt so that up to
  // D = S-1 tiles stay in flight across the barrier (never __syncthreads(): its fence drains vmcnt to 0).
  DIGA_TCLK(1);
  DIGA_TCLK(2);
#ifdef DIGA_TILE_CLOCK
  const unsigned long long sclk0 = clock64();
#endif
  // ---- fused residual input, 1x1, two K steps of operand registers in flight (small M tiles only: the register sets must fit).
  // The register-staged path above has ONE step of loads in flight, and those loads are the block's HBM reads (raw3 and the
  // residual: nothing re-reads them), so a step lasts one memory latency: 5.1 us per step on layer3.conv1 at 15 frames.  Here
  // step t issues the loads of step t + 2, and the block-input write-back of step t is issued AFTER them, so that the counted
  // wait for step t + 1's weights (LDS-DMA, issued first) leaves both the loads and the stores outstanding.
  // Issue order per step: weights DMA (t + 1), operand loads (t + 2), write-back (t); the wait at the top of step t + 1 allows
  // exactly the NA loads and, in the blocks that write back, the A_CH stores to stay in flight (the stores are issued by
  // every lane -- rows outside the sample write their zeros to the zero page -- so that their number never varies).
  constexpr bool DEEP = AMODE == 2 && BK == 64 && A_CH <= 2;
  bool deep = false;
  if constexpr (DEEP) deep = p.ntaps == 1 && p.nterms == 1 && (nt & 1) == 0 && nt >= 2;
  if constexpr (DEEP) {
    if (deep) {
#ifdef DIGA_EXP_STAGGER_US  /* experiment only: every other CU's first block starts late, so that the CUs' epilogues (HBM idle) do not coincide */
      if (blockIdx.x < 256 && ((blockIdx.x >> 3) & 1)) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < 100ll * DIGA_EXP_STAGGER_US) __builtin_amdgcn_s_sleep(32);
      }
#endif
      constexpr int NA = 4 + 2 * A_CH;               // loads of one operand set: 2 + 2 affine vectors, A_CH raw, A_CH residual chunks
      constexpr int KCH = 64 / MF, KS = BK / (8 * KCH);
      const bool wb = ntile == 0;
      struct RegSet {
        u32x4 raw[A_CH], rsd[A_CH];
        f32x4 sc[2], sh[2];
      };
      RegSet r0, r1;
      u32x4 o[A_CH];
      const float* const scp = p.a_scale + (size_t)n_img * p.a_affine_stride + cch * 8;
      const float* const shp = p.a_shift + (size_t)n_img * p.a_affine_stride + cch * 8;
      const unsigned short* const res = reinterpret_cast<const unsigned short*>(p.a_res);
      unsigned short* const aout = reinterpret_cast<unsigned short*>(p.a_out);
      // Rows beyond the sample (ih0 = -100000): their loads go to the sample's last row instead (any valid address: the
      // values are zeroed in storeA) and their write-back stores go to the zero page with a zero step -- every load and store
      // below is then ONE unconditional instruction, which is what the counted waits rely on (a select between a tensor
      // address and the zero page turns into a load of the zero page + a drain + a masked load)
      int ldoff[A_CH];
      unsigned short* wbp[A_CH];
      int wbstep[A_CH];
      unsigned okmask[A_CH];                          // all ones / zero: an AND instead of a branch around the transform
      {
        const int oh = (p.hw - 1) / p.Wo, ow = (p.hw - 1) - oh * p.Wo;
        const int lastoff = ((n_img * p.H + oh * p.stride) * p.W + ow * p.stride) * p.Cin + cch * 8;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
          const bool ok = ih0[i] >= 0;
          ldoff[i] = ok ? pixoff[i] : lastoff;
          wbp[i] = ok ? aout + pixoff[i] : const_cast<unsigned short*>(zero);
          wbstep[i] = ok ? 64 : 0;
          okmask[i] = ok ? 0xffffffffu : 0u;
        }
      }
      // The operand loads are asm statements: with compiler-visible loads in this loop hipcc guards every use with
      // s_waitcnt vmcnt(0) (it does not count across the LDS-DMA and the loop's back edge), which drains the very loads that
      // are meant to stay in flight.  Their completion is counted by wait_b below, and landed() ties the registers to that
      // wait (cdna_hip_programming.md 7, asm loads: "=v" loads, wait, "+v" tie, sched_barrier).
      auto loadA = [&](RegSet& r, int tile) {
        const int c0 = tile * 64;
        const float *sp = scp + c0, *hp = shp + c0;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.sc[0]) : "v"(sp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(r.sc[1]) : "v"(sp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.sh[0]) : "v"(hp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(r.sh[1]) : "v"(hp) : "memory");
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
          const unsigned short *rp = in + (ldoff[i] + c0), *qp = res + (ldoff[i] + c0);
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.raw[i]) : "v"(rp) : "memory");
          asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(r.rsd[i]) : "v"(qp) : "memory");   // read for the last time
        }
      };
      auto landed = [&](RegSet& r) {                  // after wait_b: the set's registers hold their data from here on
        asm volatile("" : "+v"(r.sc[0]), "+v"(r.sc[1]), "+v"(r.sh[0]), "+v"(r.sh[1]), "+v"(r.raw[0]), "+v"(r.rsd[0])::"memory");
        if constexpr (A_CH == 2) asm volatile("" : "+v"(r.raw[1]), "+v"(r.rsd[1])::"memory");
        __builtin_amdgcn_sched_barrier(0);
      };
      auto storeA = [&](const RegSet& r, int stg) {   // bn_apply_kernel<1>'s arithmetic, rounded once (as a_store above)
        unsigned char* const dA = smem + stg * STAGE_BYTES + wave * 1024 + lane * 16;
        const float sc[8] = {r.sc[0][0], r.sc[0][1], r.sc[0][2], r.sc[0][3], r.sc[1][0], r.sc[1][1], r.sc[1][2], r.sc[1][3]};
        const float sh[8] = {r.sh[0][0], r.sh[0][1], r.sh[0][2], r.sh[0][3], r.sh[1][0], r.sh[1][1], r.sh[1][2], r.sh[1][3]};
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
          float v[8], q[8];
          unpack_bf16x8(r.raw[i], v);
          unpack_bf16x8(r.rsd[i], q);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(fmaf(v[e], sc[e], sh[e]) + q[e], 0.f);
          o[i] = pack_bf16x8(v) & u32x4{okmask[i], okmask[i], okmask[i], okmask[i]};
          *reinterpret_cast<u32x4*>(dA + i * NW * 1024) = o[i];
        }
      };
      auto issueB = [&](int stg, int tile) {
        unsigned char* const dB = smem + stg * STAGE_BYTES + A_BYTES + wave * 1024;
#pragma unroll
        for (int j = 0; j < B_CH; ++j)
          __builtin_amdgcn_global_load_lds((gbl_ptr)(reinterpret_cast<const char*>(wgt + tile * 64) + wrow_off[j]), (lds_ptr)(dB + j * NW * 1024), 16, 0, 0);
      };
      auto writeback = [&]() {                       // called once per K step, in K order: the pointers walk the channels
        if (wb) {
#pragma unroll
          for (int i = 0; i < A_CH; ++i) {
            *reinterpret_cast<u32x4*>(wbp[i]) = o[i];  // (zeros onto the zero page for rows outside the sample)
            wbp[i] += wbstep[i];
          }
        }
      };
      auto wait_b = [&](bool loads) {                  // the weights of the step about to run have landed; younger ops stay in flight
        if (loads) {
          if (wb) wait_vmcnt<NA + A_CH>(); else wait_vmcnt<NA>();
        } else {
          if (wb) wait_vmcnt<A_CH>(); else wait_vmcnt<0>();
        }
      };
      auto run_stage = [&](int stg) {
        const unsigned char* bA = smem + stg * STAGE_BYTES + fa_row;
        const unsigned char* bB = smem + stg * STAGE_BYTES + fb_row;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int coff = ((KCH * s + hh) ^ fr_swz) << 4;
          bf16x8 fa[MT], fb[NT];
#pragma unroll
          for (int a = 0; a < MT; ++a) fa[a] = *reinterpret_cast<const bf16x8*>(bA + a * MF * ROWB + coff);
#pragma unroll
          for (int b = 0; b < NT; ++b) fb[b] = *reinterpret_cast<const bf16x8*>(bB + b * MF * ROWB + coff);
#pragma unroll
          for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) {
              if constexpr (MF == 32)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
              else
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
            }
        }
      };
      loadA(r0, 0);
      __builtin_amdgcn_sched_barrier(0);
      issueB(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      loadA(r1, 1);
      __builtin_amdgcn_sched_barrier(0);
      for (int t = 0; t < nt; t += 2) {
        // even step: tile t, register set 0, LDS stage 0
        if (t == 0) wait_vmcnt<NA>(); else wait_b(true);
        landed(r0);
        storeA(r0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issueB(1, t + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < nt) loadA(r0, t + 2);
        __builtin_amdgcn_sched_barrier(0);
        writeback();
        __builtin_amdgcn_sched_barrier(0);
        run_stage(0);
        // odd step: tile t + 1, register set 1, LDS stage 1
        wait_b(t + 2 < nt);
        landed(r1);
        storeA(r1, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nt) issueB(0, t + 2);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 3 < nt) loadA(r1, t + 3);
        __builtin_amdgcn_sched_barrier(0);
        writeback();
        __builtin_amdgcn_sched_barrier(0);
        run_stage(1);
      }
    }
  }
  const int nt_generic = deep ? 0 : nt;             // (the generic loop below is skipped when the deep loop ran)
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < nt_generic) issue_next(s);
  int stage = 0, fill = D % S;
  for (int t = 0; t < nt_generic; ++t) {
    // tile t must have landed: leave at most min(D-1, nt-1-t) younger tiles of THIS wave outstanding
    const int younger = min(D - 1, nt_generic - 1 - t);
    if (younger >= 2) wait_vmcnt<(D >= 3 ? 2 : 0) * LPT>();
    else if (younger == 1) wait_vmcnt<(D >= 2 ? 1 : 0) * LPT>();
    else wait_vmcnt<0>();
    if constexpr (AMODE > 0) {          // this wave's A chunks of tile t have landed in registers: transform, write them to LDS
      a_store(stage);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();       // every wave's part of tile t is in LDS; everyone is done reading tile t-1
    const bool more = t + D < nt_generic;   // K step t+D refills the buffer tile t-1 was read from
    if (!ILV && more) issue_next(fill);
    const unsigned char* bA = smem + stage * STAGE_BYTES + fa_row;
    const unsigned char* bB = smem + stage * STAGE_BYTES + fb_row;
    constexpr int KCH = 64 / MF;          // 16-byte chunks one MFMA consumes along K: 2 (K = 16) or 4 (K = 32)
    constexpr int KS = BK / (8 * KCH);
    auto ksub = [&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      const int coff = ((KCH * s + hh) ^ fr_swz) << 4;
      bf16x8 fa[MT], fb[NT];
#pragma unroll
      for (int a = 0; a < MT; ++a) fa[a] = *reinterpret_cast<const bf16x8*>(bA + a * MF * ROWB + coff);
#pragma unroll
      for (int b = 0; b < NT; ++b) fb[b] = *reinterpret_cast<const bf16x8*>(bB + b * MF * ROWB + coff);
      // this group's share of the next K step's loads, in the shadow of the MFMAs below (riding only the first half of the
      // step's groups, so that the pieces have longer to land, changed nothing: +-2 %, same-box A/B)
      if (ILV && more) issue_pieces(fill, std::integral_constant<int, s * LPT / KS>{}, std::integral_constant<int, (s + 1) * LPT / KS>{});
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          if constexpr (MF == 32)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
          else
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
    };
    ksub(std::integral_constant<int, 0>{});
    if constexpr (KS >= 2) ksub(std::integral_constant<int, 1>{});
    if constexpr (KS == 4) {
      ksub(std::integral_constant<int, 2>{});
      ksub(std::integral_constant<int, 3>{});
    }
    if (ILV && more) advance_next();
    stage = stage + 1 == S ? 0 : stage + 1;
    fill = fill + 1 == S ? 0 : fill + 1;
  }
  __syncthreads();                      // all LDS reads done: the epilogue reuses the ring
  DIGA_TCLK(3);
#ifdef DIGA_TILE_CLOCK
  const unsigned long long sclk1 = clock64();
#endif

  // ------------------------------------------------------------------ epilogue
  // accumulator element r of tile (a,b): row = wm*WTM + acc_row(a, r), col = wn*WTN + acc_col(b)
  if constexpr (STATS) {
    // one statistics unit per 32 rows (stat_group / stat_fold32 above); lanes 0-31 end up with the unit's sum, lanes 32-63 with
    // its sum of squares, for the column they own
    constexpr int UT = 32 / MF;                           // MFMA tile rows per unit: 1 or 2
#pragma unroll
    for (int ua = 0; ua < MT / UT; ++ua) {
      const int ubase = r0 + wm * WTM + ua * 32;          // first row (inside the sample) of this unit: wave-uniform
      if (ubase < p.hw) {
        const bool full = ubase + 32 <= p.hw;
        const size_t urow = ((size_t)n_img * p.units_per_sample + (ubase >> 5)) * 2;
        auto unit_stats = [&](auto full_c) {
          constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
          for (int b = 0; b < NT; ++b) {
            const int c = n0 + wn * WTN + acc_col(b);
            const float kshift = c < p.Cout ? p.shift[c] : 0.f;
            // the group held in registers 4q .. 4q+3 of tile (a, b), whose first row inside the unit is row0
            auto group = [&](int a, int q, int row0) {
              unsigned valid = 15u;
              if constexpr (!FULL) {
                const int left = p.hw - (ubase + row0);     // rows of this group inside the sample
                valid = left >= 4 ? 15u : left <= 0 ? 0u : (1u << left) - 1u;
              }
              return stat_group<FULL>(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3], kshift, valid);
            };
            float w;
            if constexpr (MF == 32) {
              const f32x2 g0 = group(ua, 0, 4 * hh), g1 = group(ua, 1, 8 + 4 * hh), g2 = group(ua, 2, 16 + 4 * hh), g3 = group(ua, 3, 24 + 4 * hh);
              w = stat_fold32((g0 + g2) + (g1 + g3));       // (g_hh + g_4+hh) + (g_2+hh + g_6+hh), then hh = 0 + hh = 1
            } else {
              const f32x2 ga = group(2 * ua, 0, 4 * hh), gb = group(2 * ua + 1, 0, 16 + 4 * hh);   // groups fq and 4 + fq
              w = xor_add16(stat_fold32(ga + gb));          // + the pair of fq ^ 2, then even groups + odd groups
            }
            if (c < p.Cout) {                               // (16x16x32: lanes fq and fq ^ 1 store the same word)
              float* const dst = &p.slab[(urow + (lane >> 5)) * p.Cout + c];
              if (p.fin_ticket)      // read back inside this launch by another block, maybe on another XCD: write through
                __hip_atomic_store(dst, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              else
                *dst = w;
            }
          }
        };
        if (full)
          unit_stats(std::true_type{});
        else
          unit_stats(std::false_type{});
      }
    }
  }

  DIGA_TCLK(4);
  if constexpr (OUT_F32) {
    float* out = reinterpret_cast<float*>(p.out) + (size_t)blockIdx.y * p.M * p.cout_pad;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const int c = n0 + wn * WTN + acc_col(b);
#pragma unroll
        for (int r = 0; r < AR; ++r) {
          const int rem = r0 + wm * WTM + acc_row(a, r);
          if (rem < p.hw) out[(size_t)(m_base + rem) * p.cout_pad + c] = acc[a][b][r];
        }
      }
  } else if constexpr (EPI != EPI_STATS_ONLY) {
    unsigned short* sC = reinterpret_cast<unsigned short*>(smem);   // [BM][SC_LD] bf16
    // 16x16x32 accumulators put rows 4 apart on the four lane groups of one ds_write_b16: 16 bytes of row padding keep
    // them two-way on the banks, as the 32x32x16 layout is
    constexpr int SC_LD = MF == 16 ? BN + 8 : BN;
    constexpr int CPR = BN / 8;                       // 16-byte chunks per tile row
    if constexpr (EPI == EPI_FUSED_APPLY) {
      // the residual tile comes in first, with coalesced 16-byte loads (read for the last time: non-temporal)
      const unsigned short* res = reinterpret_cast<const unsigned short*>(p.ep_res);
#pragma unroll
      for (int i = 0; i < BM * CPR / NTHR; ++i) {
        const int id = tid + NTHR * i;
        const int row = id / CPR, cc = id - row * CPR;
        const int rem = r0 + row, c = n0 + cc * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (rem < p.hw && c < p.Cout)
          v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(res + (size_t)(m_base + rem) * p.Cout + c));
        *reinterpret_cast<u32x4*>(sC + row * SC_LD + cc * 8) = v;
      }
      __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const int col = wn * WTN + acc_col(b);
        float esc = 1.f, esh = 0.f;
        if constexpr (EPI == EPI_FUSED_APPLY) {
          const int c = n0 + col;
          if (c < p.Cout) {
            esc = p.ep_scale[(size_t)n_img * p.ep_affine_stride + c];
            esh = p.ep_shift[(size_t)n_img * p.ep_affine_stride + c];
          }
        }
#pragma unroll
        for (int r = 0; r < AR; ++r) {
          const int row = wm * WTM + acc_row(a, r);
          if constexpr (EPI == EPI_FUSED_APPLY) {
            // the arithmetic of bn_apply_kernel<1> on the unrounded accumulator: y = fma(x, scale, shift) + residual
            float y = fmaf(acc[a][b][r], esc, esh) + bf16_bits_to_f32(sC[row * SC_LD + col]);
            if (p.ep_relu) y = fmaxf(y, 0.f);
            sC[row * SC_LD + col] = f32_to_bf16_bits(y);
          } else {
            sC[row * SC_LD + col] = f32_to_bf16_bits(acc[a][b][r]);
          }
        }
      }
    __syncthreads();
    DIGA_TCLK(5);
    unsigned short* out = reinterpret_cast<unsigned short*>(p.out);
#pragma unroll
    for (int i = 0; i < BM * CPR / NTHR; ++i) {
      const int id = tid + NTHR * i;
      const int row = id / CPR, cc = id - row * CPR;
      const int rem = r0 + row, c = n0 + cc * 8;
      if (rem < p.hw && c < p.Cout)
        // the raw output is read exactly once, by bn_apply: a non-temporal store keeps it from displacing the A / B lines
        // the other tiles still need in L2 (+2.5 % whole-step throughput, same-box A/B)
        __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(sC + row * SC_LD + cc * 8), reinterpret_cast<u32x4*>(out + (size_t)(m_base + rem) * p.Cout + c));
    }
  }

  DIGA_TCLK(6);
#ifdef DIGA_TILE_CLOCK
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DIGA_TCLK(7);
  if (tid == 0 && blockIdx.x < 16384)
  {
    for (int i = 0; i < 8; ++i) g_tile_clock[(size_t)blockIdx.x * 16 + i] = tclk[i];
    g_tile_clock[(size_t)blockIdx.x * 16 + 10] = sclk1 - sclk0;     // shader-clock cycles of the K loop
  }
#endif
  if constexpr (STATS) {
    if (p.fin_ticket) {
      // ---- fused DAM finalize.  gfx942 / gfx950 SPECIFIC memory ordering (this library is built for gfx950 only): the hand-off
      // below is not the HIP memory model's release / acquire pair -- a release fence writes the whole L2 back and an acquire
      // invalidates it, see the measurement below -- it relies on (a) agent-scope atomic stores being write-through to memory
      // on these parts, (b) s_waitcnt vmcnt(0) returning only once those stores are performed at the memory side, (c) agent-scope
      // atomic loads bypassing the reader's L2.  The XCDs' L2s are not coherent with each other inside a launch: this block's slab values went
      // out as agent-scope write-through stores, it takes a ticket for its (sample, N tile) group, and the block that draws the
      // last ticket reduces the group with agent-scope (L2-bypassing) loads: the arithmetic and the summation order are those
      // of bn_finalize_kernel (dam_affine.h), so scale / shift are bit-identical to the two-launch form.
      // (the slab stores above are agent-scope write-through stores when the finalize is fused: wait until they are performed.
      // A release FENCE here writes the whole L2 back and an acquire fence invalidates it -- the L2 the other tiles' operand
      // re-use lives on: measured 296 vs 708 frames/s.)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                     // every thread's slab stores are released; the LDS staging tile is dead
      int* s_last = reinterpret_cast<int*>(smem);
      if (tid == 0) {
        unsigned* tk = p.fin_ticket + n_img * p.ntiles_n + ntile;
        const unsigned prev = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev == (unsigned)p.tiles_per_sample - 1u;
        if (last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        *s_last = last;
      }
      __syncthreads();
      const int last = *s_last;
      __syncthreads();                     // s_last is read before the reduction overwrites the LDS
      if (last) {
        constexpr int PARTS = NTHR / BN;                        // threads per channel: 1..16, a power of two
        static_assert(PARTS >= 1 && PARTS <= FIN_PH && FIN_PH % PARTS == 0, "phases are dealt evenly over a channel's threads");
        double* red = reinterpret_cast<double*>(smem);          // [FIN_PH][2][BN]
        const int u0 = n_img * p.units_per_sample, u1 = u0 + p.units_per_sample;
        {
          const int cl = tid % BN, part = tid / BN, c = n0 + cl;
          for (int ph = part; ph < FIN_PH; ph += PARTS) {
            double sm[2] = {0, 0};
            if (c < p.Cout) slab_range_sums<true>(p.slab, u0, u1, p.Cout, c, ph, sm);
            red[(ph * 2 + 0) * BN + cl] = sm[0];
            red[(ph * 2 + 1) * BN + cl] = sm[1];
          }
        }
        __syncthreads();
        for (int cl = tid; cl < BN; cl += NTHR) {
          const int c = n0 + cl;
          if (c >= p.Cout) continue;
          double sm[4] = {0, 0, 0, 0};
          for (int w = 0; w < FIN_PH; ++w) {
            sm[0] += red[(w * 2 + 0) * BN + cl];
            sm[1] += red[(w * 2 + 1) * BN + cl];
          }
          dam_affine_from_sums(sm, c, p.fin_n, 0.0, p.shift, p.fin_rvar, p.fin_weight, p.fin_bias, p.fin_lam, p.fin_eps,
                               p.fin_scale + (size_t)n_img * p.Cout, p.fin_shift + (size_t)n_img * p.Cout);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weights-stationary 1x1 convolution for short K (Cin = 64 / 128 / 256, stride 1: the bottlenecks' conv3), tile id 91.
//
// The tiled kernel above spends most of a K = 256 tile outside its K loop (prologue latency, statistics, staging, stores:
// 532 TFLOP/s on layer3.conv3 at 15 frames).  Here a block keeps its 128-column slice of the weights in LDS for its whole
// life (whole K: <= 64 KiB, swizzled as above) and its 8 waves run INDEPENDENTLY, no block barrier after the weights have
// landed: a wave takes one statistics unit (32 rows of one sample) at a time, streams the unit's A fragments global ->
// registers (16 KiB in flight per wave, the next unit's loads issued as this unit's registers retire), multiplies against
// the resident weights, takes the unit's statistics, stages its 32 x 128 outputs through its own 8.5 KiB of LDS and stores
// 256-byte row pieces.  Waves drift apart, so one wave's epilogue runs under the MFMAs of the wave it shares a SIMD with.
// The K order of every output element (32-channel MFMA steps, ascending) and the statistics tree are those of the tiled
// kernel: outputs and slab are bit-identical (tests/test_gpu_kernels.py).
// Grid = (unit groups) x (Cout / 128) blocks; the slices of one unit group get consecutive logical ids = one XCD, whose L2
// serves the re-reads of the group's A rows.
// Tile id 92 = the same kernel with 64-column slices for K = 512 (layer4's conv3, layer3's downsample): the slice is again 64 KiB
// of LDS, a wave's A fragments are 128 registers and its MFMA work per unit (128) equals that of a K = 256, 128-column unit, while
// the epilogue -- statistics, staging and stores are what a unit's time is made of -- covers half as many accumulators.
// NP > 1 (tile id 94): a unit's BN columns are taken in NP passes of BN / NP columns over the SAME A registers -- half the accumulators
// and weight fragments per pass, so that NW = 12 waves (three per SIMD instead of two, <= 168 registers) share the resident slice.
template <int KC, int BN, int NP = 1, int NW = 8>
__global__ __launch_bounds__(64 * NW) void conv1x1_ws_kernel(const ConvKParams p, int units_total, int units_per_group) {
  constexpr int BNP = BN / NP;                                  // columns per pass
  constexpr int NTW = BN / 16, NTP = BNP / 16, KS = KC / 32;
  constexpr int LPR = BNP / 8, RPS = 64 / LPR, NST = 32 / RPS;  // output stores: lanes per row, rows per store instruction, stores per pass
  constexpr int SQ = BNP / 4;                                   // statistics read-back: lanes per row of the pass's [2][BNP] words
  constexpr int ROWB = KC * 2, B_BYTES = BN * ROWB;
  constexpr int SW_LD = BNP + 8;                               // a wave's staging tile: 32 rows, 16 bytes of padding per row,
  constexpr int SW_BYTES = 32 * SW_LD * 2 + 2 * BNP * 4;       // + the pass's 2 x BNP statistics words (one 16-byte store per lane)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef DIGA_TILE_CLOCK
  const unsigned long long wk0 = wall_clock64();
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, fq = lane >> 4;
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int nslices = p.Cout / BN;
  const int grp = bid / nslices, n0 = (bid - grp * nslices) * BN;
  const int u_begin = grp * units_per_group, u_end = min(u_begin + units_per_group, units_total);

  // ---- the weight slice -> LDS, whole K: row r (output channel n0 + r) = ROWB bytes, 16-byte chunk c stored in slot c ^ f(r)
  // (f = r & 15, or (r >> 1) & 7 for 128-byte rows): a ds_read_b128 of 16 consecutive rows at one chunk is conflict-free
  auto bswz = [](int row) { return KC == 64 ? (row >> 1) & 7 : row & 15; };
  {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    const unsigned short* wgt = reinterpret_cast<const unsigned short*>(p.wgt);
    for (int pc = wave; pc < B_BYTES / 1024; pc += NW) {          // 1 KiB pieces, wave-uniform trip count
      const int off = pc * 1024 + lane * 16;
      const int row = off / ROWB, slot = (off % ROWB) >> 4;
      __builtin_amdgcn_global_load_lds((gbl_ptr)(wgt + (size_t)(n0 + row) * KC + ((slot ^ bswz(row)) << 3)),
                                       (lds_ptr)(smem + pc * 1024), 16, 0, 0);
    }
  }
  float ksh[NTW];
#pragma unroll
  for (int n = 0; n < NTW; ++n) ksh[n] = p.shift[n0 + n * 16 + l15];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const unsigned short* const in = reinterpret_cast<const unsigned short*>(p.in);
  unsigned short* const out = reinterpret_cast<unsigned short*>(p.out);
  unsigned short* const sW = reinterpret_cast<unsigned short*>(smem + B_BYTES + wave * SW_BYTES);
  float* const sS = reinterpret_cast<float*>(smem + B_BYTES + wave * SW_BYTES + 32 * SW_LD * 2);
  const unsigned char* const bB = smem + l15 * ROWB;
  const int b_swz = bswz(l15);                       // column tiles start at multiples of 16 rows: the swizzle depends on l15 only

  // A fragments of one unit: M tile a (16 rows), K step k: lane (l15, fq) holds channels 32k + 8fq .. + 7 of row 16a + l15
  u32x4 fa[KS][2];
  auto unit_rows = [&](int u, const unsigned short** rp) {
    const int n_img = u / p.units_per_sample, ubase = (u - n_img * p.units_per_sample) * 32;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int row = min(ubase + a * 16 + l15, p.hw - 1);      // rows beyond the sample: any valid address, never used
      rp[a] = in + ((size_t)n_img * p.hw + row) * KC + fq * 8;
    }
  };
  int u = u_begin + wave;
  if (u < u_end) {
    const unsigned short* rp[2];
    unit_rows(u, rp);
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[k][a] = *reinterpret_cast<const u32x4*>(rp[a] + k * 32);
  }
#ifdef DIGA_WS_STAGGER
  // waves w and w + NW / 2 share a SIMD (a workgroup's waves go round the SIMDs): started together they stay in phase -- both in the
  // K loop, then both in the epilogue -- and nothing overlaps; the second half starts half a unit late
  if (wave >= NW / 2) __builtin_amdgcn_s_sleep(DIGA_WS_STAGGER);
#endif
#ifdef DIGA_TILE_CLOCK
  unsigned long long wc[4] = {0, 0, 0, 0}, wt0 = clock64(), wr0 = wall_clock64();
#define DIGA_WCLK(i, t0) wc[i] += clock64() - (t0)
#else
#define DIGA_WCLK(i, t0) \
  do {                  \
  } while (0)
#endif
  for (; u < u_end; u += NW) {
    const int un = u + NW < u_end ? u + NW : u;       // the last unit re-reads itself: no branch in the body
    const unsigned short* rp[2];
    unit_rows(un, rp);
    const int n_img = u / p.units_per_sample, ubase = (u - n_img * p.units_per_sample) * 32;
    const bool full = ubase + 32 <= p.hw;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
#ifdef DIGA_TILE_CLOCK
    const unsigned long long c0 = clock64();
#endif
    f32x4 acc[2][NTP];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int n = 0; n < NTP; ++n) acc[a][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (plain loads: the compiler counts them.  Issuing and counting them by hand -- vmcnt(2 (KS - 1) + stores per unit), so
    // that a K step never waits for the previous unit's stores -- was built and measured: no gain, 105 vs 107 us)
#ifdef DIGA_EXP_WS_NOB   /* experiment only (profiles/experiments/ws_ablate): the weight fragments are read from LDS once per unit */
    bf16x8 fb[NTP];
#endif
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int coff = ((4 * k + fq) ^ b_swz) << 4;
#ifdef DIGA_EXP_WS_NOB
      if (k == 0)
#else
      bf16x8 fb[NTP];
#endif
#pragma unroll
      for (int n = 0; n < NTP; ++n) fb[n] = *reinterpret_cast<const bf16x8*>(bB + (ps * NTP + n) * 16 * ROWB + coff);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const bf16x8 av = __builtin_bit_cast(bf16x8, fa[k][a]);
#pragma unroll
        for (int n = 0; n < NTP; ++n) acc[a][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, fb[n], acc[a][n], 0, 0, 0);
      }
      // this K step's registers are free: the next unit's fragments go into them while the remaining steps run
#ifndef DIGA_EXP_WS_NOA  /* experiment only: without it the first unit's A fragments are re-used for every unit */
      if (ps == NP - 1) {
#pragma unroll
        for (int a = 0; a < 2; ++a) fa[k][a] = *reinterpret_cast<const u32x4*>(rp[a] + k * 32);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    DIGA_WCLK(0, c0);
#ifdef DIGA_TILE_CLOCK
    const unsigned long long c1 = clock64();
#endif
    // ---- the unit's statistics (the tree of the tiled kernel) and its 32 x 128 outputs
    auto unit_stats = [&](auto full_c) {
      constexpr bool FULL = decltype(full_c)::value;
      unsigned va = 15u, vb = 15u;                     // rows of this lane's two groups inside the sample (ragged last unit)
      if constexpr (!FULL) {
        const int la = p.hw - (ubase + 4 * fq), lb = la - 16;
        va = la >= 4 ? 15u : la <= 0 ? 0u : (1u << la) - 1u;
        vb = lb >= 4 ? 15u : lb <= 0 ? 0u : (1u << lb) - 1u;
      }
#pragma unroll
      for (int n = 0; n < NTP; ++n) {
        const f32x2 ga = stat_group<FULL>(acc[0][n][0], acc[0][n][1], acc[0][n][2], acc[0][n][3], ksh[ps * NTP + n], va);
        const f32x2 gb = stat_group<FULL>(acc[1][n][0], acc[1][n][1], acc[1][n][2], acc[1][n][3], ksh[ps * NTP + n], vb);
        const float w = xor_add16(stat_fold32(ga + gb));
        // lanes 0-31 hold the unit's sum, lanes 32-63 its sum of squares (lanes fq and fq ^ 1 the same word): collected in
        // LDS and stored below with ONE 16-byte store per lane -- store instructions, not bytes, are what an epilogue pays for
        sS[(lane >> 5) * BNP + n * 16 + l15] = w;
      }
    };
#ifdef DIGA_EXP_WS_NOEPI   /* experiments only (profiles/experiments/ws_ablate): parts of the epilogue left out, results wrong */
    {
      float t = 0.f;
#pragma unroll
      for (int n = 0; n < NTP; ++n) t += acc[0][n][0] + acc[1][n][3];
      if (t == 123.456f) sS[lane] = t;      // keeps the MFMAs alive
      continue;
    }
#endif
#ifndef DIGA_EXP_WS_NOSTAT
    if (full)
      unit_stats(std::true_type{});
    else
      unit_stats(std::false_type{});
#endif
    DIGA_WCLK(3, c1);
#ifdef DIGA_EXP_WS_NOSTAGE
    {
      float t = 0.f;
#pragma unroll
      for (int n = 0; n < NTP; ++n) t += acc[0][n][0] + acc[1][n][3] + acc[0][n][1] + acc[1][n][2];
      if (t == 123.456f) sW[lane] = (unsigned short)lane;
    }
#else
#pragma unroll
    for (int n = 0; n < NTP; ++n) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
          const unsigned pk = pack_bf16x2(acc[a][n][i], acc[a][n][i + 1]);      // rows i, i + 1 of the same column
          sW[(a * 16 + 4 * fq + i) * SW_LD + n * 16 + l15] = (unsigned short)pk;
          sW[(a * 16 + 4 * fq + i + 1) * SW_LD + n * 16 + l15] = (unsigned short)(pk >> 16);
        }
    }
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's staging writes are done (LDS ops of one wave stay in order)
    __builtin_amdgcn_wave_barrier();
    DIGA_WCLK(1, c1);
#ifdef DIGA_TILE_CLOCK
    const unsigned long long c2 = clock64();
#endif
    const size_t orow = (size_t)n_img * p.hw + ubase;
    u32x4 ov[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) ov[i] = *reinterpret_cast<const u32x4*>(sW + (i * RPS + lane / LPR) * SW_LD + (lane % LPR) * 8);
    if (lane < 2 * SQ) {                                                       // row lane / SQ, columns 4 (lane % SQ) .. + 3
      const f32x4 sv = *reinterpret_cast<const f32x4*>(sS + lane * 4);
      *reinterpret_cast<f32x4*>(&p.slab[((size_t)u * 2 + lane / SQ) * p.Cout + n0 + ps * BNP + (lane % SQ) * 4]) = sv;
    }
    unsigned short* const orp = out + (orow + lane / LPR) * p.Cout + n0 + ps * BNP + (lane % LPR) * 8;
    if (full) {
#pragma unroll
      for (int i = 0; i < NST; ++i) __builtin_nontemporal_store(ov[i], reinterpret_cast<u32x4*>(orp + (size_t)i * RPS * p.Cout));
    } else {
#pragma unroll
      for (int i = 0; i < NST; ++i)
        if (ubase + i * RPS + lane / LPR < p.hw) __builtin_nontemporal_store(ov[i], reinterpret_cast<u32x4*>(orp + (size_t)i * RPS * p.Cout));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the staging tile has been read: the next unit may overwrite it
    __builtin_amdgcn_wave_barrier();
    DIGA_WCLK(2, c2);
    }  // passes
  }
#ifdef DIGA_TILE_CLOCK
  if (lane == 0 && blockIdx.x < 2048) {
    unsigned long long* g = g_tile_clock + ((size_t)blockIdx.x * NW + wave) * 8;
    g[0] = wc[0], g[1] = wc[1], g[2] = wc[2], g[3] = clock64() - wt0, g[4] = wall_clock64() - wr0;
    g[5] = (u_end - u_begin - wave + NW - 1) / NW;
    g[6] = wc[3];
    g[7] = wr0 - wk0;
  }
#endif
}

int launch_ws(const diga_conv_desc* d, const ConvKParams& p, int epi, int id, hipStream_t st) {
  const int bn = id == 92 ? 64 : 128;                      // columns of the resident slice
  const int np = id == 94 ? 2 : 1, nw = id == 94 ? 12 : 8;   // passes per unit, waves per block
  DIGA_CHECK_ARG(d->ntaps == 1 && d->stride == 1 && d->dy[0] == 0 && d->dx[0] == 0 && d->H == d->Ho && d->W == d->Wo,
                 "diga_conv_igemm_bf16: tile %d (weights-stationary) is for 1x1 stride-1 convolutions", id);
  if (id != 92)
    DIGA_CHECK_ARG((d->Cin == 64 || d->Cin == 128 || d->Cin == 256) && d->Cout % 128 == 0 && d->cout_pad == d->Cout,
                   "diga_conv_igemm_bf16: tile %d needs Cin in {64, 128, 256} and Cout %% 128 == 0 (got %d -> %d)", id, d->Cin, d->Cout);
  else
    DIGA_CHECK_ARG(d->Cin == 512 && d->Cout % 64 == 0 && d->cout_pad == d->Cout,
                   "diga_conv_igemm_bf16: tile 92 needs Cin = 512 and Cout %% 64 == 0 (got %d -> %d)", d->Cin, d->Cout);
  DIGA_CHECK_ARG(epi == EPI_BF16_STATS && p.nterms == 1 && d->splitk == 1 && p.fin_ticket == nullptr,
                 "diga_conv_igemm_bf16: tile %d has the trunk epilogue only (bf16 output + statistics)", id);
  const int units_total = d->B * p.units_per_sample, nslices = d->Cout / bn;
  // one block per CU (its LDS): as many unit groups as fill the chip once, at least one unit per wave and group
  int groups = 256 / nslices;
  if (groups < 1) groups = 1;
  if (groups > cdiv(units_total, nw)) groups = cdiv(units_total, nw);
  const int upg = cdiv(units_total, groups);
  groups = cdiv(units_total, upg);
  const int bnp = bn / np;
  const int smem = bn * d->Cin * 2 + nw * (32 * (bnp + 8) * 2 + 2 * bnp * 4);
  dim3 grid(groups * nslices), block(64 * nw);
#define DIGA_WS_LAUNCH(KC, BN, NP, NW)                                                                                  \
  {                                                                                                                     \
    static std::once_flag once;                                                                                         \
    std::call_once(once, [&] {                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_ws_kernel<KC, BN, NP, NW>),                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize,                                             \
                                BN * KC * 2 + NW * (32 * (BN / NP + 8) * 2 + 2 * (BN / NP) * 4));                       \
    });                                                                                                                 \
    hipLaunchKernelGGL((conv1x1_ws_kernel<KC, BN, NP, NW>), grid, block, smem, st, p, units_total, upg);                \
  }
  if (id == 94) {
    if (d->Cin == 64) DIGA_WS_LAUNCH(64, 128, 2, 12)
    else if (d->Cin == 128) DIGA_WS_LAUNCH(128, 128, 2, 12)
    else DIGA_WS_LAUNCH(256, 128, 2, 12)
  } else if (d->Cin == 64) DIGA_WS_LAUNCH(64, 128, 1, 8)
  else if (d->Cin == 128) DIGA_WS_LAUNCH(128, 128, 1, 8)
  else if (d->Cin == 256) DIGA_WS_LAUNCH(256, 128, 1, 8)
  else DIGA_WS_LAUNCH(512, 64, 1, 8)
#undef DIGA_WS_LAUNCH
  return DIGA_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Weights-stationary 1x1 convolution, software-pipelined inside the wave (tile id 95).
//
// The ablations of tile 91 (profiles/experiments/ws_ablate) say its parts ADD -- a wave's K loop (waits for its LDS / global
// reads) and its epilogue (a chain of ~400 dependent vector / LDS / store instructions) run one after the other, 8 450 cycles
// per unit for 2 048 cycles of MFMA, and two waves per SIMD are not enough to cover each other.  Here a wave overlaps them
// ITSELF: a unit's 128 columns are taken as two halves of 64 (two accumulator sets of 32 registers over the same A registers),
// and while the MFMAs of one half run, the epilogue of the PREVIOUS half -- statistics and staging of one 16-column tile per K
// step, then the read-back, then the stores -- is issued between them.  Same K order, same statistics tree, same rounding:
// bit-identical to tiles 91 / 23 (tests/test_gpu_kernels.py).
template <int KC, int NW>
__global__ __launch_bounds__(64 * NW) void conv1x1_wsp_kernel(const ConvKParams p, int units_total, int units_per_group) {
  constexpr int BN = 128, BNP = 64, NTW = BN / 16, NTP = BNP / 16, KS = KC / 32;
  constexpr int LPR = BNP / 8, RPS = 64 / LPR, NST = 32 / RPS;   // output stores of one half: lanes per row, rows per store, stores
  constexpr int SQ = BNP / 4;
  constexpr int ROWB = KC * 2, B_BYTES = BN * ROWB;
  constexpr int SW_LD = BNP + 8;
  constexpr int SW_BYTES = 32 * SW_LD * 2 + 2 * BNP * 4;
  constexpr int PPS = 8 / KS;                                    // epilogue parts per K step (8 parts: 4 tiles, read-back, stores, -, -)
  static_assert(KS == 8 || KS == 4 || KS == 2, "K = 256 / 128 / 64");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, fq = lane >> 4;
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int nslices = p.Cout / BN;
  const int grp = bid / nslices, n0 = (bid - grp * nslices) * BN;
  const int u_begin = grp * units_per_group, u_end = min(u_begin + units_per_group, units_total);
  auto bswz = [](int row) { return KC == 64 ? (row >> 1) & 7 : row & 15; };
  {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    const unsigned short* wgt = reinterpret_cast<const unsigned short*>(p.wgt);
    for (int pc = wave; pc < B_BYTES / 1024; pc += NW) {
      const int off = pc * 1024 + lane * 16;
      const int row = off / ROWB, slot = (off % ROWB) >> 4;
      __builtin_amdgcn_global_load_lds((gbl_ptr)(wgt + (size_t)(n0 + row) * KC + ((slot ^ bswz(row)) << 3)),
                                       (lds_ptr)(smem + pc * 1024), 16, 0, 0);
    }
  }
  float ksh[NTW];
#pragma unroll
  for (int n = 0; n < NTW; ++n) ksh[n] = p.shift[n0 + n * 16 + l15];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const unsigned short* const in = reinterpret_cast<const unsigned short*>(p.in);
  unsigned short* const out = reinterpret_cast<unsigned short*>(p.out);
  unsigned short* const sW = reinterpret_cast<unsigned short*>(smem + B_BYTES + wave * SW_BYTES);
  float* const sS = reinterpret_cast<float*>(smem + B_BYTES + wave * SW_BYTES + 32 * SW_LD * 2);
  const unsigned char* const bB = smem + l15 * ROWB;
  const int b_swz = bswz(l15);

  u32x4 fa[KS][2];
  auto unit_rows = [&](int u, const unsigned short** rp) {
    const int n_img = u / p.units_per_sample, ubase = (u - n_img * p.units_per_sample) * 32;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int row = min(ubase + a * 16 + l15, p.hw - 1);
      rp[a] = in + ((size_t)n_img * p.hw + row) * KC + fq * 8;
    }
  };
  int u = u_begin + wave;
  if (u < u_end) {
    const unsigned short* rp[2];
    unit_rows(u, rp);
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[k][a] = *reinterpret_cast<const u32x4*>(rp[a] + k * 32);
  }

  // ---- one part of the epilogue of a finished half (its accumulators `ac`, unit uE = (n_img, ubase), half HE): parts 0..3 = the
  // statistics and the staging writes of 16-column tile `part`; 4 = read-back and stores.  FULL (the unit lies inside its sample)
  // is a template argument chosen ONCE per half, so that a K step with its epilogue part is one basic block the scheduler can
  // interleave
  auto epi_part = [&](auto part_c, auto he_c, auto full_c, f32x4 (&ac)[2][NTP], int uE, int n_img, int ubase) {
    constexpr int part = decltype(part_c)::value, HE = decltype(he_c)::value;
    constexpr bool FULL = decltype(full_c)::value;
    if constexpr (part < NTP) {
      constexpr int n = part;
      unsigned va = 15u, vb = 15u;
      if constexpr (!FULL) {
        const int la = p.hw - (ubase + 4 * fq), lb = la - 16;
        va = la >= 4 ? 15u : la <= 0 ? 0u : (1u << la) - 1u;
        vb = lb >= 4 ? 15u : lb <= 0 ? 0u : (1u << lb) - 1u;
      }
      const f32x2 ga = stat_group<FULL>(ac[0][n][0], ac[0][n][1], ac[0][n][2], ac[0][n][3], ksh[HE * NTP + n], va);
      const f32x2 gb = stat_group<FULL>(ac[1][n][0], ac[1][n][1], ac[1][n][2], ac[1][n][3], ksh[HE * NTP + n], vb);
      const float w = xor_add16(stat_fold32(ga + gb));
      sS[(lane >> 5) * BNP + n * 16 + l15] = w;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
          const unsigned pk = pack_bf16x2(ac[a][n][i], ac[a][n][i + 1]);
          sW[(a * 16 + 4 * fq + i) * SW_LD + n * 16 + l15] = (unsigned short)pk;
          sW[(a * 16 + 4 * fq + i + 1) * SW_LD + n * 16 + l15] = (unsigned short)(pk >> 16);
        }
    } else if constexpr (part == NTP) {
      // (LDS ops of one wave stay in order: the reads below see this wave's staging writes)
      u32x4 ov[NST];
#pragma unroll
      for (int i = 0; i < NST; ++i) ov[i] = *reinterpret_cast<const u32x4*>(sW + (i * RPS + lane / LPR) * SW_LD + (lane % LPR) * 8);
      const f32x4 sv = *reinterpret_cast<const f32x4*>(sS + (lane & (2 * SQ - 1)) * 4);
      const size_t orow = (size_t)n_img * p.hw + ubase;
      if (lane < 2 * SQ)
        *reinterpret_cast<f32x4*>(&p.slab[((size_t)uE * 2 + lane / SQ) * p.Cout + n0 + HE * BNP + (lane % SQ) * 4]) = sv;
      unsigned short* const orp = out + (orow + lane / LPR) * p.Cout + n0 + HE * BNP + (lane % LPR) * 8;
      if constexpr (FULL) {
#pragma unroll
        for (int i = 0; i < NST; ++i) __builtin_nontemporal_store(ov[i], reinterpret_cast<u32x4*>(orp + (size_t)i * RPS * p.Cout));
      } else {
#pragma unroll
        for (int i = 0; i < NST; ++i)
          if (ubase + i * RPS + lane / LPR < p.hw) __builtin_nontemporal_store(ov[i], reinterpret_cast<u32x4*>(orp + (size_t)i * RPS * p.Cout));
      }
    }
  };

  // ---- one half: the MFMAs of half H of the current unit into `am`; EPI: with the epilogue of the pending half (`ae`, unit uE,
  // half 1 - H) issued between the K steps.  No branch inside: one basic block per half, the
  // scheduler interleaves the epilogue's vector instructions with the MFMAs and hoists the next step's LDS reads.
  auto half = [&](auto h_c, auto epi_c, auto full_c, f32x4 (&am)[2][NTP], f32x4 (&ae)[2][NTP], int uE, const unsigned short* const* rp) {
    constexpr int H = decltype(h_c)::value;
    constexpr bool EPI = decltype(epi_c)::value;
    int n_img = 0, ubase = 0;
    if constexpr (EPI) {
      n_img = uE / p.units_per_sample;
      ubase = (uE - n_img * p.units_per_sample) * 32;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int n = 0; n < NTP; ++n) am[a][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    // GUIDED (the 4-wave form: one wave per SIMD has nobody to hide its waits, and the registers for it): the weight fragments of K
    // step k + 1 are read at the top of step k into the other half of fb, and the step is laid out by sched_group_barrier as
    // [4 LDS reads] 8 x [1 MFMA, 6 vector, 1 LDS write] [the rest] -- left alone the scheduler reads them right before their MFMAs
    constexpr bool GUIDED = NW == 4;
    bf16x8 fb[2][NTP];
    if constexpr (GUIDED) {
#pragma unroll
      for (int n = 0; n < NTP; ++n) fb[0][n] = *reinterpret_cast<const bf16x8*>(bB + (H * NTP + n) * 16 * ROWB + ((fq ^ b_swz) << 4));
    }
    static_for<KS>([&](auto k_c) {
      constexpr int k = decltype(k_c)::value;
      constexpr int cur = GUIDED ? (k & 1) : 0;
      if constexpr (GUIDED) {
        if constexpr (k + 1 < KS) {
          const int coff = ((4 * (k + 1) + fq) ^ b_swz) << 4;
#pragma unroll
          for (int n = 0; n < NTP; ++n) fb[(k + 1) & 1][n] = *reinterpret_cast<const bf16x8*>(bB + (H * NTP + n) * 16 * ROWB + coff);
        }
      } else {
        const int coff = ((4 * k + fq) ^ b_swz) << 4;
#pragma unroll
        for (int n = 0; n < NTP; ++n) fb[0][n] = *reinterpret_cast<const bf16x8*>(bB + (H * NTP + n) * 16 * ROWB + coff);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const bf16x8 av = __builtin_bit_cast(bf16x8, fa[k][a]);
#pragma unroll
        for (int n = 0; n < NTP; ++n) am[a][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, fb[cur][n], am[a][n], 0, 0, 0);
      }
      if constexpr (H == 1) {      // the unit's last use of this K step's A registers: the next unit's fragments go into them
#pragma unroll
        for (int a = 0; a < 2; ++a) fa[k][a] = *reinterpret_cast<const u32x4*>(rp[a] + k * 32);
      }
      if constexpr (EPI)
        static_for<PPS>([&](auto q_c) {
          epi_part(std::integral_constant<int, k * PPS + decltype(q_c)::value>{}, std::integral_constant<int, 1 - H>{}, full_c, ae, uE, n_img, ubase);
        });
      if constexpr (GUIDED) {
        if constexpr (k + 1 < KS) __builtin_amdgcn_sched_group_barrier(0x100, NTP, 0);
#pragma unroll
        for (int i = 0; i < 2 * NTP; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if constexpr (EPI) {
            __builtin_amdgcn_sched_group_barrier(0x002, 6 * PPS, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 2 * PPS, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      } else {
        __builtin_amdgcn_sched_barrier(0x100);    // only LDS reads (the next step's weight fragments) may move across K steps
      }
    });
  };
  auto is_full = [&](int uE) {
    const int n_img = uE / p.units_per_sample, ubase = (uE - n_img * p.units_per_sample) * 32;
    return ubase + 32 <= p.hw;
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;

  f32x4 acc0[2][NTP], acc1[2][NTP];
  bool pend = false;
  int uprev = u;
  for (; u < u_end; u += NW) {
    const int un = u + NW < u_end ? u + NW : u;
    const unsigned short* rp[2];
    unit_rows(un, rp);
    if (!pend)
      half(C0{}, std::false_type{}, std::true_type{}, acc0, acc1, uprev, rp);
    else if (is_full(uprev))
      half(C0{}, std::true_type{}, std::true_type{}, acc0, acc1, uprev, rp);
    else
      half(C0{}, std::true_type{}, std::false_type{}, acc0, acc1, uprev, rp);
    if (is_full(u))
      half(C1{}, std::true_type{}, std::true_type{}, acc1, acc0, u, rp);
    else
      half(C1{}, std::true_type{}, std::false_type{}, acc1, acc0, u, rp);
    pend = true;
    uprev = u;
  }
  if (pend) {
    const int n_img = uprev / p.units_per_sample, ubase = (uprev - n_img * p.units_per_sample) * 32;
    static_for<NTP + 1>([&](auto part_c) {
      if (ubase + 32 <= p.hw)
        epi_part(part_c, C1{}, std::true_type{}, acc1, uprev, n_img, ubase);
      else
        epi_part(part_c, C1{}, std::false_type{}, acc1, uprev, n_img, ubase);
    });
  }
}

int launch_wsp(const diga_conv_desc* d, const ConvKParams& p, int epi, int id, hipStream_t st) {
  const int nw = id == 96 ? 4 : 8;       // 96: one wave per SIMD (up to 512 registers: nothing spills at K = 256)
  DIGA_CHECK_ARG(d->ntaps == 1 && d->stride == 1 && d->dy[0] == 0 && d->dx[0] == 0 && d->H == d->Ho && d->W == d->Wo,
                 "diga_conv_igemm_bf16: tile %d (weights-stationary, pipelined) is for 1x1 stride-1 convolutions", id);
  DIGA_CHECK_ARG((d->Cin == 64 || d->Cin == 128 || d->Cin == 256) && d->Cout % 128 == 0 && d->cout_pad == d->Cout,
                 "diga_conv_igemm_bf16: tile %d needs Cin in {64, 128, 256} and Cout %% 128 == 0 (got %d -> %d)", id, d->Cin, d->Cout);
  DIGA_CHECK_ARG(epi == EPI_BF16_STATS && p.nterms == 1 && d->splitk == 1 && p.fin_ticket == nullptr,
                 "diga_conv_igemm_bf16: tile %d has the trunk epilogue only (bf16 output + statistics)", id);
  const int units_total = d->B * p.units_per_sample, nslices = d->Cout / 128;
  int groups = 256 / nslices;
  if (groups < 1) groups = 1;
  if (groups > cdiv(units_total, nw)) groups = cdiv(units_total, nw);
  const int upg = cdiv(units_total, groups);
  groups = cdiv(units_total, upg);
  const int smem = 128 * d->Cin * 2 + nw * (32 * (64 + 8) * 2 + 2 * 64 * 4);
  dim3 grid(groups * nslices), block(64 * nw);
#define DIGA_WSP_LAUNCH(KC, NW)                                                                                         \
  {                                                                                                                     \
    static std::once_flag once;                                                                                         \
    std::call_once(once, [&] {                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_wsp_kernel<KC, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                128 * KC * 2 + NW * (32 * (64 + 8) * 2 + 2 * 64 * 4));                                                  \
    });                                                                                                                 \
    hipLaunchKernelGGL((conv1x1_wsp_kernel<KC, NW>), grid, block, smem, st, p, units_total, upg);                       \
  }
  if (nw == 8) {
    if (d->Cin == 64) DIGA_WSP_LAUNCH(64, 8)
    else if (d->Cin == 128) DIGA_WSP_LAUNCH(128, 8)
    else DIGA_WSP_LAUNCH(256, 8)
  } else {
    if (d->Cin == 64) DIGA_WSP_LAUNCH(64, 4)
    else if (d->Cin == 128) DIGA_WSP_LAUNCH(128, 4)
    else DIGA_WSP_LAUNCH(256, 4)
  }
#undef DIGA_WSP_LAUNCH
  return DIGA_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// A-stationary 1x1 convolution with the DAM apply of its INPUT fused (tile id 93): conv3 of a bottleneck taking conv2's RAW
// output, A = bf16(relu(fma(raw2, scale, shift))) formed ONCE per element in registers -- act2 is never written and read back
// (VERDICT r2 item 7).  The fused-input forms above repeat the transform for every N tile that re-reads an element (4-8x
// on a conv3: measured loss); here a wave keeps one statistics unit (32 rows x K) of transformed A in registers and the
// WEIGHTS stream past it: 64-column slices through a 3-deep LDS-DMA ring shared by the block's 8 waves, one raw barrier per
// slice.  K order (32-channel MFMA steps, ascending), statistics tree, transform arithmetic and rounding are those of diga_bn_apply +
// the tiled kernel: output and slab are bit-identical (tests/test_gpu_kernels.py).
// MEASURED (round 3, layer3's conv3 at 15 frames): 135 us against 22 + 95 us for diga_bn_apply + the weights-stationary kernel, so
// the engine does NOT use it.  Phase clock: an interval's per-wave chain is 1 450 cycles of MFMA phase (64 MFMAs = 1 024 issue cycles)
// + 2 300 of epilogue for its 32 accumulator registers + ~600 of barrier / DMA issue, and the block's lockstep lets nothing overlap
// it; the weights-stationary kernel spends 8 450 cycles on twice the columns with its waves drifting freely.  Two units per wave would
// halve the fixed share but need 128 + 64 + 32 registers at K = 256.
// vmcnt: loads, stores and LDS-DMA count together in issue order (MI355X_MICROARCH.md); a wave issues the pieces of slice q + 2
// FIRST in interval q, so at the top of interval q + 1 "all but the PCS youngest operations" covers slice q + 1 whatever number of
// stores the interval issued (never fewer operations behind slice q + 1 than the PCS pieces of slice q + 2; vmcnt(0) at the tail).
template <int KC>
__global__ __launch_bounds__(512) void conv1x1_as_kernel(const ConvKParams p, int units_total, int ngroups) {
  constexpr int BN = 64, NTW = BN / 16, KS = KC / 32, NSTG = 3;
  constexpr int ROWB = KC * 2, STAGE = BN * ROWB, PCS = STAGE / 8192;
  constexpr int SW_LD = BN + 8, SW_BYTES = 32 * SW_LD * 2 + 2 * BN * 4;
  constexpr int LPR = BN / 8, RPS = 64 / LPR, NST = 32 / RPS, SQ = BN / 4;
  static_assert(PCS >= 1 && STAGE % 8192 == 0, "a stage is a whole number of 8-wave LDS-DMA rounds");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // stagger: waves 4-7 run their epilogue one interval behind (epilogue of slice q - 1, then the MFMAs of slice q) so that on every
  // SIMD one wave's statistics / staging / stores run beside the other's MFMAs.  Built, bit-identical, and SLOWER (148 vs 135 us on
  // layer3's conv3 at 15 frames: the late waves' epilogue, contended by their partners' MFMA phase, takes 3 800 cycles instead of
  // 2 300 and the early waves sit 2 800 cycles at the barrier -- phase clock, profiles/experiments/as_conv3_r3.log): off
  constexpr bool kStagger = false;
  const bool late = kStagger && wave >= 4;
  const int l15 = lane & 15, fq = lane >> 4;
  const int NS = p.Cout / BN;                         // weight slices
  const int nit = (ngroups - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // unit groups of this block
  const int Q = nit * NS;                             // intervals
  auto bswz = [](int row) { return KC == 64 ? (row >> 1) & 7 : row & 15; };
  const unsigned short* const wgt = reinterpret_cast<const unsigned short*>(p.wgt);
  const unsigned short* const in = reinterpret_cast<const unsigned short*>(p.in);
  unsigned short* const out = reinterpret_cast<unsigned short*>(p.out);
  // this lane's source offset inside a slice, per piece: row / slot of the lane-linear LDS image (swizzle on the source side)
  int woff[PCS];
#pragma unroll
  for (int i = 0; i < PCS; ++i) {
    const int off = (wave + 8 * i) * 1024 + lane * 16;
    const int row = off / ROWB, slot = (off % ROWB) >> 4;
    woff[i] = row * KC + ((slot ^ bswz(row)) << 3);
  }
  auto issue_slice = [&](int sl, int buf) {
    const unsigned short* src = wgt + (size_t)sl * BN * KC;
#pragma unroll
    for (int i = 0; i < PCS; ++i)
      __builtin_amdgcn_global_load_lds((gbl_ptr)(src + woff[i]), (lds_ptr)(smem + buf * STAGE + (wave + 8 * i) * 1024), 16, 0, 0);
  };
  unsigned short* const sW = reinterpret_cast<unsigned short*>(smem + NSTG * STAGE + wave * SW_BYTES);
  float* const sS = reinterpret_cast<float*>(smem + NSTG * STAGE + wave * SW_BYTES + 32 * SW_LD * 2);
  // the statistics shift of every output channel, once, in LDS: a register load of it inside the slice loop would make the
  // compiler drain the vector-memory counter -- and with it the weight slices in flight -- in every epilogue
  float* const sK = reinterpret_cast<float*>(smem + NSTG * STAGE + 8 * SW_BYTES);
  for (int c = tid; c < p.Cout; c += 512) sK[c] = p.shift[c];
  const int b_swz = bswz(l15);

  u32x4 fa[KS][2];
  f32x4 acc[2][NTW];
  // this wave's copy of its unit's DAM affine (scale [KC] | shift [KC]): two 16-byte loads per lane together with the unit's rows,
  // then LDS reads per K step (loading the 8 + 8 values of every K step from memory took one round trip per step)
  float* const sAf = reinterpret_cast<float*>(smem + NSTG * STAGE + 8 * SW_BYTES + 4 * KC * 4 + wave * 2 * KC * 4);
  f32x4 afs = {0.f, 0.f, 0.f, 0.f}, afh = {0.f, 0.f, 0.f, 0.f};
  auto load_unit = [&](int u) {                        // raw conv2 output of unit u: M tile a, K step k: channels 32k + 8fq .. + 7
    const int n_img = u / p.units_per_sample, ubase = (u - n_img * p.units_per_sample) * 32;
    if (lane * 4 < KC) {
      afs = *reinterpret_cast<const f32x4*>(p.a_scale + (size_t)n_img * p.a_affine_stride + lane * 4);
      afh = *reinterpret_cast<const f32x4*>(p.a_shift + (size_t)n_img * p.a_affine_stride + lane * 4);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int row = min(ubase + a * 16 + l15, p.hw - 1);         // rows beyond the sample: any valid address, never used
      const unsigned short* rp = in + ((size_t)n_img * p.hw + row) * KC + fq * 8;
#pragma unroll
      for (int k = 0; k < KS; ++k) fa[k][a] = *reinterpret_cast<const u32x4*>(rp + k * 32);
    }
  };
  auto transform_unit = [&]() {                        // bn_apply_kernel<0>'s arithmetic and its single rounding
    if (lane * 4 < KC) {
      *reinterpret_cast<f32x4*>(sAf + lane * 4) = afs;
      *reinterpret_cast<f32x4*>(sAf + KC + lane * 4) = afh;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const float* sp = sAf + k * 32 + fq * 8;
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(sp + KC), h1 = *reinterpret_cast<const f32x4*>(sp + KC + 4);
      const float sc[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
      const float sh[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        float v[8];
        unpack_bf16x8(fa[k][a], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(fmaf(v[e], sc[e], sh[e]), 0.f);
        fa[k][a] = pack_bf16x8(v);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the affine copy has been read: the next unit may overwrite it)
    __builtin_amdgcn_wave_barrier();
  };
  auto mfma_slice = [&](int buf) {
    const unsigned char* const bB = smem + buf * STAGE + l15 * ROWB;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int n = 0; n < NTW; ++n) acc[a][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int coff = ((4 * k + fq) ^ b_swz) << 4;
      bf16x8 fb[NTW];
#pragma unroll
      for (int n = 0; n < NTW; ++n) fb[n] = *reinterpret_cast<const bf16x8*>(bB + n * 16 * ROWB + coff);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const bf16x8 av = __builtin_bit_cast(bf16x8, fa[k][a]);
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[a][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, fb[n], acc[a][n], 0, 0, 0);
      }
    }
  };
  // statistics + staged stores of unit u, columns [n0, n0 + 64): the epilogue of conv1x1_ws_kernel
  auto epilogue = [&](int u, int n0) {
    const int n_img = u / p.units_per_sample, ubase = (u - n_img * p.units_per_sample) * 32;
    const bool full = ubase + 32 <= p.hw;
    float ksh[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) ksh[n] = sK[n0 + n * 16 + l15];
    auto unit_stats = [&](auto full_c) {
      constexpr bool FULL = decltype(full_c)::value;
      unsigned va = 15u, vb = 15u;
      if constexpr (!FULL) {
        const int la = p.hw - (ubase + 4 * fq), lb = la - 16;
        va = la >= 4 ? 15u : la <= 0 ? 0u : (1u << la) - 1u;
        vb = lb >= 4 ? 15u : lb <= 0 ? 0u : (1u << lb) - 1u;
      }
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        const f32x2 ga = stat_group<FULL>(acc[0][n][0], acc[0][n][1], acc[0][n][2], acc[0][n][3], ksh[n], va);
        const f32x2 gb = stat_group<FULL>(acc[1][n][0], acc[1][n][1], acc[1][n][2], acc[1][n][3], ksh[n], vb);
        sS[(lane >> 5) * BN + n * 16 + l15] = xor_add16(stat_fold32(ga + gb));
      }
    };
    if (full)
      unit_stats(std::true_type{});
    else
      unit_stats(std::false_type{});
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
          const unsigned pk = pack_bf16x2(acc[a][n][i], acc[a][n][i + 1]);
          sW[(a * 16 + 4 * fq + i) * SW_LD + n * 16 + l15] = (unsigned short)pk;
          sW[(a * 16 + 4 * fq + i + 1) * SW_LD + n * 16 + l15] = (unsigned short)(pk >> 16);
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // EXACTLY NST + 1 stores per epilogue (the interval waits below count them): rows beyond the sample re-store the sample's last
    // row with that row's own values -- the same bytes to the same address, harmless -- instead of being skipped
    const int rlast = p.hw - 1 - ubase;               // last row of the unit inside the sample (>= 31 for a full unit)
    u32x4 ov[NST];
    int orow[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      orow[i] = min(i * RPS + lane / LPR, rlast);
      ov[i] = *reinterpret_cast<const u32x4*>(sW + orow[i] * SW_LD + (lane % LPR) * 8);
    }
    {
      const int sl_ = lane < 2 * SQ ? lane : lane - 2 * SQ;      // (lanes 32-63 repeat lanes 0-31: one unconditional store)
      const f32x4 sv = *reinterpret_cast<const f32x4*>(sS + sl_ * 4);
      *reinterpret_cast<f32x4*>(&p.slab[((size_t)u * 2 + sl_ / SQ) * p.Cout + n0 + (sl_ % SQ) * 4]) = sv;
    }
    unsigned short* const orp = out + ((size_t)n_img * p.hw + ubase) * p.Cout + n0 + (lane % LPR) * 8;
#pragma unroll
    for (int i = 0; i < NST; ++i) __builtin_nontemporal_store(ov[i], reinterpret_cast<u32x4*>(orp + (size_t)orow[i] * p.Cout));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  constexpr int EST = NST + 1;                         // stores of one epilogue

  // prologue: slices 0 and 1 on their way; the shift table is complete before anyone's first epilogue (the loop's first barrier)
  issue_slice(0, 0);
  if (Q > 1) issue_slice(1 % NS, 1);
  int buf = 0, sl = 0, grp = blockIdx.x;               // ring buffer / slice / unit group of interval q
  int u = grp * 8 + wave;
  int pu = 0, pn0 = 0;                                 // (unit, first column) whose accumulators a late wave still holds
  bool pending = false;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this wave's sK writes)
  // Operations this wave issued AFTER the pieces of the slice it is about to wait for, oldest first: the stores of the interval
  // those pieces were issued in (st1: 0 or EST), the pieces of the following slice (PCS, none at the tail), the stores of the interval
  // just finished (st0).  "All but that many youngest operations are done" is exactly "the slice has landed" -- a wait that also
  // covered the stores would put a store's round trip into every interval (measured: 4.5 us per interval instead of ~2).
  int st0 = 0, st1 = 0;
#ifdef DIGA_TILE_CLOCK
  unsigned long long ck[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ct = clock64();
#define AS_CLK(i)                        \
  {                                      \
    const unsigned long long t_ = clock64(); \
    ck[i] += t_ - ct;                    \
    ct = t_;                             \
  }
#else
#define AS_CLK(i) \
  do {            \
  } while (0)
#endif
  for (int q = 0; q < Q; ++q) {
    {
      const int younger = st1 + st0 + (q > 0 && q + 1 < Q ? PCS : 0);       // (q = 0: both prologue slices are still in front)
      if (q == 0) { if (Q > 1) wait_vmcnt<PCS>(); else wait_vmcnt<0>(); }
      else if (younger == 2 * EST + PCS) wait_vmcnt<2 * EST + PCS>();
      else if (younger == EST + PCS) wait_vmcnt<EST + PCS>();
      else if (younger == PCS) wait_vmcnt<PCS>();
      else if (younger == 2 * EST) wait_vmcnt<2 * EST>();
      else if (younger == EST) wait_vmcnt<EST>();
      else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();                                  // ... everyone's; and everyone is done reading slice q - 1
    AS_CLK(0);
    if (q + 2 < Q) issue_slice((sl + 2) % NS, (buf + 2) % NSTG);   // refills the buffer slice q - 1 was read from
    __builtin_amdgcn_sched_barrier(0);
    const bool live = u < units_total;
    st1 = st0;
    st0 = (late ? pending : live) ? EST : 0;            // the epilogue this interval runs
    // one body for both wave groups: [late: epilogue of the previous interval] [MFMAs of this slice] [early: its epilogue]
    AS_CLK(1);
    if (sl == 0 && live) load_unit(u);                 // (late waves: in flight under the epilogue below)
    AS_CLK(2);
    if (late && pending) epilogue(pu, pn0);
    AS_CLK(3);
    if (sl == 0 && live) transform_unit();
    AS_CLK(4);
    if (live) mfma_slice(buf);
#ifdef DIGA_TILE_CLOCK
    asm volatile("s_nop 0" : "+v"(acc[0][0]), "+v"(acc[1][NTW - 1]));   // (the stamp below follows the MFMAs' results)
#endif
    AS_CLK(5);
    if (late) {
      pending = live;
      pu = u, pn0 = sl * BN;
    } else if (live) {
      epilogue(u, sl * BN);
    }
    AS_CLK(6);
    buf = buf + 1 == NSTG ? 0 : buf + 1;
    if (++sl == NS) {
      sl = 0;
      grp += gridDim.x;
      u = grp * 8 + wave;
    }
  }
  if (late && pending) epilogue(pu, pn0);
#ifdef DIGA_TILE_CLOCK
  if (lane == 0 && blockIdx.x < 2048) {
    unsigned long long* g = g_tile_clock + ((size_t)blockIdx.x * 8 + wave) * 8;
    for (int i = 0; i < 7; ++i) g[i] = ck[i];
    g[7] = Q;
  }
#endif
#undef AS_CLK
}

int launch_as(const diga_conv_desc* d, const ConvKParams& p, int epi, int amode, hipStream_t st) {
  DIGA_CHECK_ARG(amode == 1, "diga_conv_igemm_bf16_fused_in: tile 93 (A-stationary) takes the plain fused input (no residual); it has no unfused form");
  DIGA_CHECK_ARG(d->ntaps == 1 && d->stride == 1 && d->dy[0] == 0 && d->dx[0] == 0 && d->H == d->Ho && d->W == d->Wo,
                 "diga_conv_igemm_bf16_fused_in: tile 93 (A-stationary) is for 1x1 stride-1 convolutions");
  DIGA_CHECK_ARG((d->Cin == 64 || d->Cin == 128 || d->Cin == 256) && d->Cout % 64 == 0 && d->Cout <= 4 * d->Cin && d->cout_pad == d->Cout,
                 "diga_conv_igemm_bf16_fused_in: tile 93 needs Cin in {64, 128, 256}, Cout %% 64 == 0 and Cout <= 4 Cin (got %d -> %d)", d->Cin, d->Cout);
  DIGA_CHECK_ARG(epi == EPI_BF16_STATS && p.nterms == 1 && d->splitk == 1 && p.fin_ticket == nullptr,
                 "diga_conv_igemm_bf16_fused_in: tile 93 has the trunk epilogue only (bf16 output + statistics)");
  const int units_total = d->B * p.units_per_sample, ngroups = cdiv(units_total, 8);
  const int blocks = ngroups < 256 ? ngroups : 256;          // persistent: one block per CU walks the unit groups
  const int smem = 3 * 64 * d->Cin * 2 + 8 * (32 * (64 + 8) * 2 + 2 * 64 * 4) + 4 * d->Cin * 4 + 8 * 2 * d->Cin * 4;
  DIGA_CHECK_ARG(smem <= 160 * 1024, "diga_conv_igemm_bf16_fused_in: tile 93: Cout %d needs %d bytes of LDS", d->Cout, smem);
#define DIGA_AS_LAUNCH(KC)                                                                                              \
  {                                                                                                                     \
    static std::once_flag once;                                                                                         \
    std::call_once(once, [&] {                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_as_kernel<KC>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                3 * 64 * KC * 2 + 8 * (32 * (64 + 8) * 2 + 2 * 64 * 4) + 4 * KC * 4 + 8 * 2 * KC * 4);  \
    });                                                                                                                 \
    hipLaunchKernelGGL(conv1x1_as_kernel<KC>, dim3(blocks), dim3(512), smem, st, p, units_total, ngroups);              \
  }
  if (d->Cin == 64) DIGA_AS_LAUNCH(64)
  else if (d->Cin == 128) DIGA_AS_LAUNCH(128)
  else DIGA_AS_LAUNCH(256)
#undef DIGA_AS_LAUNCH
  return DIGA_OK;
}

struct TileCfg {
  int id, BM, BN, S, BK = 64;
  bool ilv = false;
};
// id = shape + 10*(4 - ring depth).  shape 1: 128x128, 2: 128x64, 3: 64x64, 4: 64x128, 5: 128x32 (head); 8-wave blocks:
// 6: 256x128, 7: 128x256, 8: 256x256 (ring depth 2 only).  ids 31..38: the same shapes with 32-channel K steps and a 4-deep
// ring (same LDS as 64-channel steps with 2 buffers, but three stages of loads in flight).
// LDS per block = S*(BM+BN)*128 B, so the ring depth also sets how many blocks share a CU (160 KiB).
const TileCfg kTiles[] = {{1, 128, 128, 4}, {11, 128, 128, 3}, {21, 128, 128, 2}, {2, 128, 64, 4}, {12, 128, 64, 3}, {22, 128, 64, 2},
                          {3, 64, 64, 4},   {13, 64, 64, 3},   {23, 64, 64, 2},   {4, 64, 128, 4}, {14, 64, 128, 3}, {24, 64, 128, 2},
                          {5, 128, 32, 4},  {15, 128, 32, 3},  {25, 128, 32, 2},  {16, 256, 128, 3}, {26, 256, 128, 2}, {17, 128, 256, 3},
                          {27, 128, 256, 2}, {28, 256, 256, 2},
                          {31, 128, 128, 4, 32}, {32, 128, 64, 4, 32}, {33, 64, 64, 4, 32}, {34, 64, 128, 4, 32}, {36, 256, 128, 4, 32},
                          {37, 128, 256, 4, 32}, {38, 256, 256, 4, 32},
                          // ids 46..48 / 56..58: ring depth 2 (64-channel steps) / 4 (32-channel steps) with the next step's
                          // loads interleaved with the MFMA groups
                          // loads interleaved with the MFMA groups (pays for the 8-wave tiles only; 4-wave 256x256 tiles -- one
                          // wave per SIMD, 256 accumulator registers -- were measured 15-25 % slower and are not built)
                          {46, 256, 128, 2, 64, true}, {47, 128, 256, 2, 64, true}, {48, 256, 256, 2, 64, true},
                          {56, 256, 128, 4, 32, true}, {57, 128, 256, 4, 32, true}, {58, 256, 256, 4, 32, true},
                          // ids 6x: 32-channel steps with only 2 buffers -- half the LDS, so that two (four) blocks share a CU and
                          // one block's epilogue overlaps another's K loop (the short-K 1x1 convs: K = 256 is 4-8 steps per tile)
                          {61, 128, 128, 2, 32, false}, {64, 64, 128, 2, 32, false}, {66, 256, 128, 2, 32, true}, {67, 128, 256, 2, 32, true},
                          // ids 76 / 77: the shapes of 66 / 67 without interleaved issue -- the fused-input kernels only
                          {76, 256, 128, 2, 32, false}, {77, 128, 256, 2, 32, false},
                          // id 91: conv1x1_ws_kernel (weights-stationary 1x1, Cin <= 256); BM = its 32-row units; id 92: Cin = 512, 64-column slices
                          // id 93: conv1x1_as_kernel (A-stationary 1x1 with the fused DAM apply of its input; fused-input entry point only)
                          {91, 32, 128, 1, 64, false}, {92, 32, 64, 1, 64, false}, {93, 32, 64, 3, 64, false},
                          {94, 32, 128, 1, 64, false}, {95, 32, 128, 1, 64, false}, {96, 32, 128, 1, 64, false}};

const TileCfg* find_tile(int id) {
  for (const auto& t : kTiles)
    if (t.id == id) return &t;
  return nullptr;
}

int auto_tile(const diga_conv_desc* d) {
  const int64_t M = (int64_t)d->B * d->Ho * d->Wo;
  // Measured on MI355X over the net's real shapes (tests/bench_conv.py, profiles/r1_conv_tile_sweep.log): at
  // M = 8385..33153 rows the loop is bound by what one CU can pull from L2 (~60-70 GB/s), not by MFMA issue, and
  // several small co-resident blocks per CU (2-deep ring, 32-48 KB LDS) beat one big block with a deep ring.
  if (d->cout_pad % 64 != 0) return 25;
  if (d->cout_pad % 128 == 0 && (d->cout_pad >= 512 || M >= 16000)) return 24;   // 64x128
  return 23;                                                                      // 64x64
}

template <int BM, int BN, int WM, int WN, int S, int BK = 64, bool ILV = false>
int launch_cfg(const ConvKParams& p, int splitk, int epi, hipStream_t st) {
  constexpr int main_bytes = S * (BM + BN) * BK * 2;
  constexpr int epi_bytes = BM * (DIGA_MF == 16 ? BN + 8 : BN) * 2;
  constexpr int fin_bytes = FIN_PH * 2 * BN * 8;         // the fused finalize's fp64 phase sums
  constexpr int smem01 = main_bytes > epi_bytes ? main_bytes : epi_bytes;
  constexpr int smem = smem01 > fin_bytes ? smem01 : fin_bytes;
  dim3 grid(p.mtiles * p.ntiles_n, splitk), block(64 * WM * WN);
#define DIGA_EPI_KERNEL(E) conv_igemm_kernel<BM, BN, WM, WN, S, BK, ILV, E>
  static std::once_flag attr_once;   // > 64 KiB of dynamic LDS needs the opt-in once per kernel instance (any thread may get here first)
  std::call_once(attr_once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&DIGA_EPI_KERNEL(EPI_BF16_STATS)), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&DIGA_EPI_KERNEL(EPI_F32)), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&DIGA_EPI_KERNEL(EPI_BF16)), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&DIGA_EPI_KERNEL(EPI_STATS_ONLY)), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&DIGA_EPI_KERNEL(EPI_FUSED_APPLY)), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&DIGA_EPI_KERNEL(EPI_F32_STATS)), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  });
  switch (epi) {
    case EPI_BF16_STATS: hipLaunchKernelGGL((DIGA_EPI_KERNEL(EPI_BF16_STATS)), grid, block, smem, st, p); break;
    case EPI_F32: hipLaunchKernelGGL((DIGA_EPI_KERNEL(EPI_F32)), grid, block, smem, st, p); break;
    case EPI_BF16: hipLaunchKernelGGL((DIGA_EPI_KERNEL(EPI_BF16)), grid, block, smem, st, p); break;
    case EPI_STATS_ONLY: hipLaunchKernelGGL((DIGA_EPI_KERNEL(EPI_STATS_ONLY)), grid, block, smem, st, p); break;
    case EPI_FUSED_APPLY: hipLaunchKernelGGL((DIGA_EPI_KERNEL(EPI_FUSED_APPLY)), grid, block, smem, st, p); break;
    case EPI_F32_STATS: hipLaunchKernelGGL((DIGA_EPI_KERNEL(EPI_F32_STATS)), grid, block, smem, st, p); break;
    default: diga_set_error("diga_conv_igemm_bf16: bad epilogue mode %d", epi); return DIGA_EINVAL;
  }
#undef DIGA_EPI_KERNEL
  return DIGA_OK;
}

// The fused-input kernels (AMODE 1 / 2): trunk epilogue only, 2-deep ring, no interleaved issue.
template <int BM, int BN, int WM, int WN, int BK>
int launch_fused_in(const ConvKParams& p, int amode, hipStream_t st) {
  constexpr int main_bytes = 2 * (BM + BN) * BK * 2;
  constexpr int epi_bytes = BM * (DIGA_MF == 16 ? BN + 8 : BN) * 2;
  constexpr int smem = main_bytes > epi_bytes ? main_bytes : epi_bytes;
  dim3 grid(p.mtiles * p.ntiles_n, 1), block(64 * WM * WN);
#define DIGA_FUSED_KERNEL(A) conv_igemm_kernel<BM, BN, WM, WN, 2, BK, false, EPI_BF16_STATS, A>
  static std::once_flag attr_once;
  std::call_once(attr_once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&DIGA_FUSED_KERNEL(1)), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&DIGA_FUSED_KERNEL(2)), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  });
  if (amode == 1)
    hipLaunchKernelGGL((DIGA_FUSED_KERNEL(1)), grid, block, smem, st, p);
  else
    hipLaunchKernelGGL((DIGA_FUSED_KERNEL(2)), grid, block, smem, st, p);
#undef DIGA_FUSED_KERNEL
  return DIGA_OK;
}

int validate(const diga_conv_desc* d, const char* who) {
  DIGA_CHECK_ARG(d != nullptr, "%s: desc is NULL", who);
  DIGA_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0, "%s: bad spatial dims", who);
  DIGA_CHECK_ARG(d->Cin > 0 && d->Cin % 64 == 0, "%s: Cin=%d must be a positive multiple of 64", who, d->Cin);
  DIGA_CHECK_ARG(d->Cout > 0 && d->cout_pad >= d->Cout && d->cout_pad % 32 == 0, "%s: bad Cout/cout_pad %d/%d", who,
                 d->Cout, d->cout_pad);
  DIGA_CHECK_ARG(d->stride == 1 || d->stride == 2, "%s: stride %d", who, d->stride);
  DIGA_CHECK_ARG(d->ntaps >= 1 && d->ntaps <= DIGA_MAX_TAPS, "%s: ntaps %d", who, d->ntaps);
  DIGA_CHECK_ARG((d->Ho - 1) * d->stride < d->H && (d->Wo - 1) * d->stride < d->W, "%s: output grid exceeds input", who);
  DIGA_CHECK_ARG(d->splitk >= 1 && (d->splitk == 1 || d->out_f32), "%s: splitk needs out_f32", who);
  DIGA_CHECK_ARG(d->out_f32 || d->Cout % 8 == 0, "%s: bf16 output needs Cout %% 8 == 0", who);
  DIGA_CHECK_ARG(d->nplanes >= 0 && d->nplanes <= 3, "%s: nplanes %d (0/1 = bf16 operands, 2/3 = split-operand verification mode)", who, d->nplanes);
  DIGA_CHECK_ARG(d->nplanes <= 1 || (d->out_f32 && d->splitk == 1), "%s: operand planes need out_f32 and splitk == 1", who);
  const int npl = d->nplanes > 1 ? d->nplanes : 1;
  const int64_t in_bytes = (int64_t)npl * d->B * d->H * d->W * d->Cin * 2;
  const int64_t w_bytes = (int64_t)npl * d->cout_pad * d->ntaps * d->Cin * 2;
  DIGA_CHECK_ARG(in_bytes < 0x7ffffff0ll && w_bytes < 0x7ffffff0ll, "%s: tensor exceeds the 2 GiB buffer range", who);
  const int64_t M = (int64_t)d->B * d->Ho * d->Wo;
  DIGA_CHECK_ARG(M * d->cout_pad < 0x7fffffffll, "%s: output too large", who);
  return DIGA_OK;
}

}  // namespace

extern "C" int diga_conv_stat_units(const diga_conv_desc* d) {
  if (validate(d, "diga_conv_stat_units") != DIGA_OK) return DIGA_EINVAL;
  return d->B * cdiv(d->Ho * d->Wo, 32);
}

namespace {
struct FusedIn {               // fused DAM apply on the input operand (diga_conv_igemm_bf16_fused_in); amode == 0: off
  int amode = 0;
  const float *scale = nullptr, *shift = nullptr;
  const void* res = nullptr;
  void* act_out = nullptr;
  int affine_stride = 0;
};

struct FinArgs {               // fused DAM finalize (diga_conv_igemm_bf16_dam); ticket == nullptr: off
  unsigned* ticket = nullptr;
  const float *rvar = nullptr, *weight = nullptr, *bias = nullptr;
  float *scale = nullptr, *shift = nullptr;
  float lam = 0.f, eps = 0.f;
};

int run_conv(const diga_conv_desc* d, const void* in, const void* wgt, void* out, float* stat_slab, const float* stat_shift,
             int epi, const float* ep_scale, const float* ep_shift, const void* ep_res, int ep_affine_stride, int ep_relu,
             void* stream, const FinArgs& fin = FinArgs(), const FusedIn& fi = FusedIn()) {
  const TileCfg* t = find_tile(d->tile ? d->tile : auto_tile(d));
  DIGA_CHECK_ARG(t != nullptr, "diga_conv_igemm_bf16: unknown tile id %d", d->tile);
  DIGA_CHECK_ARG(d->cout_pad % t->BN == 0, "diga_conv_igemm_bf16: cout_pad %d not a multiple of tile N %d", d->cout_pad, t->BN);
  ConvKParams p;
  p.in = in;
  p.wgt = wgt;
  p.out = out;
  p.slab = stat_slab;
  p.shift = stat_shift;
  p.ep_scale = ep_scale, p.ep_shift = ep_shift, p.ep_res = ep_res, p.ep_affine_stride = ep_affine_stride, p.ep_relu = ep_relu;
  p.a_scale = fi.scale, p.a_shift = fi.shift, p.a_res = fi.res, p.a_out = fi.act_out, p.a_affine_stride = fi.affine_stride;
  p.fin_ticket = fin.ticket, p.fin_rvar = fin.rvar, p.fin_weight = fin.weight, p.fin_bias = fin.bias;
  p.fin_scale = fin.scale, p.fin_shift = fin.shift, p.fin_lam = fin.lam, p.fin_eps = fin.eps;
  p.fin_n = (double)d->Ho * d->Wo;
  p.H = d->H, p.W = d->W, p.Cin = d->Cin, p.Ho = d->Ho, p.Wo = d->Wo, p.Cout = d->Cout;
  p.stride = d->stride, p.ntaps = d->ntaps;
  p.M = d->B * d->Ho * d->Wo;
  p.K = d->ntaps * d->Cin;
  p.cout_pad = d->cout_pad;
  p.hw = d->Ho * d->Wo;
  p.tiles_per_sample = cdiv(p.hw, t->BM);
  p.units_per_sample = cdiv(p.hw, 32);
  // operand-plane products, smallest first: planes (a, b) with a + b < nplanes
  static const int kTerms2[3] = {1 | (0 << 8), 0 | (1 << 8), 0};
  static const int kTerms3[6] = {2 | (0 << 8), 0 | (2 << 8), 1 | (1 << 8), 1 | (0 << 8), 0 | (1 << 8), 0};
  p.nterms = d->nplanes == 2 ? 3 : d->nplanes == 3 ? 6 : 1;
  for (int i = 0; i < 6; ++i) p.term_ab[i] = d->nplanes == 2 ? kTerms2[i % 3] : d->nplanes == 3 ? kTerms3[i] : 0;
  p.in_plane_stride = d->B * d->H * d->W * d->Cin;
  p.wgt_plane_stride = d->cout_pad * p.K;
  p.nk_plane = p.K / t->BK;
  p.nk_total = p.nterms * p.nk_plane;
  p.nk_per_split = cdiv(p.nk_total, d->splitk);
  p.mtiles = d->B * p.tiles_per_sample;
  p.ntiles_n = d->cout_pad / t->BN;
  for (int i = 0; i < DIGA_MAX_TAPS; ++i) {
    const int dy = i < d->ntaps ? d->dy[i] : 0, dx = i < d->ntaps ? d->dx[i] : 0;
    p.dydx[i] = (int)(((unsigned)dy << 16) | ((unsigned)dx & 0xffffu));
  }
  hipStream_t st = (hipStream_t)stream;
  int e = DIGA_OK;
  if (fi.amode && t->id == 93) {
    if (int e93 = launch_as(d, p, epi, fi.amode, st)) return e93;
    DIGA_CHECK_LAUNCH("diga_conv_igemm_bf16_fused_in");
    return DIGA_OK;
  }
  if (fi.amode) {
    switch (t->id) {
      case 21: e = launch_fused_in<128, 128, 2, 2, 64>(p, fi.amode, st); break;
      case 22: e = launch_fused_in<128, 64, 2, 2, 64>(p, fi.amode, st); break;
      case 23: e = launch_fused_in<64, 64, 2, 2, 64>(p, fi.amode, st); break;
      case 24: e = launch_fused_in<64, 128, 2, 2, 64>(p, fi.amode, st); break;
      case 26: e = launch_fused_in<256, 128, 4, 2, 64>(p, fi.amode, st); break;
      case 27: e = launch_fused_in<128, 256, 2, 4, 64>(p, fi.amode, st); break;
      case 28: e = launch_fused_in<256, 256, 2, 4, 64>(p, fi.amode, st); break;
      case 61: e = launch_fused_in<128, 128, 2, 2, 32>(p, fi.amode, st); break;
      case 64: e = launch_fused_in<64, 128, 2, 2, 32>(p, fi.amode, st); break;
      case 76: e = launch_fused_in<256, 128, 4, 2, 32>(p, fi.amode, st); break;
      case 77: e = launch_fused_in<128, 256, 2, 4, 32>(p, fi.amode, st); break;
      default: diga_set_error("diga_conv_igemm_bf16_fused_in: tile id %d has no fused-input kernel", t->id); return DIGA_EINVAL;
    }
    if (e) return e;
    DIGA_CHECK_LAUNCH("diga_conv_igemm_bf16_fused_in");
    return DIGA_OK;
  }
  if (t->id == 95 || t->id == 96) {
    if (int e95 = launch_wsp(d, p, epi, t->id, st)) return e95;
    DIGA_CHECK_LAUNCH("diga_conv_igemm_bf16");
    return DIGA_OK;
  }
  if (t->id == 91 || t->id == 92 || t->id == 94) {
    if (int e91 = launch_ws(d, p, epi, t->id, st)) return e91;
    DIGA_CHECK_LAUNCH("diga_conv_igemm_bf16");
    return DIGA_OK;
  }
#define DIGA_TILE_CASES(base, BM, BN, WM, WN)                                                   \
  case base: e = launch_cfg<BM, BN, WM, WN, 4>(p, d->splitk, epi, st); break;          \
  case base + 10: e = launch_cfg<BM, BN, WM, WN, 3>(p, d->splitk, epi, st); break;     \
  case base + 20: e = launch_cfg<BM, BN, WM, WN, 2>(p, d->splitk, epi, st); break;
  switch (t->id) {
    DIGA_TILE_CASES(1, 128, 128, 2, 2)
    DIGA_TILE_CASES(2, 128, 64, 2, 2)
    DIGA_TILE_CASES(3, 64, 64, 2, 2)
    DIGA_TILE_CASES(4, 64, 128, 2, 2)
    DIGA_TILE_CASES(5, 128, 32, 4, 1)
    case 16: e = launch_cfg<256, 128, 4, 2, 3>(p, d->splitk, epi, st); break;
    case 26: e = launch_cfg<256, 128, 4, 2, 2>(p, d->splitk, epi, st); break;
    case 17: e = launch_cfg<128, 256, 2, 4, 3>(p, d->splitk, epi, st); break;
    case 27: e = launch_cfg<128, 256, 2, 4, 2>(p, d->splitk, epi, st); break;
    case 28: e = launch_cfg<256, 256, 2, 4, 2>(p, d->splitk, epi, st); break;
    case 31: e = launch_cfg<128, 128, 2, 2, 4, 32>(p, d->splitk, epi, st); break;
    case 32: e = launch_cfg<128, 64, 2, 2, 4, 32>(p, d->splitk, epi, st); break;
    case 33: e = launch_cfg<64, 64, 2, 2, 4, 32>(p, d->splitk, epi, st); break;
    case 34: e = launch_cfg<64, 128, 2, 2, 4, 32>(p, d->splitk, epi, st); break;
    case 36: e = launch_cfg<256, 128, 4, 2, 4, 32>(p, d->splitk, epi, st); break;
    case 37: e = launch_cfg<128, 256, 2, 4, 4, 32>(p, d->splitk, epi, st); break;
    case 38: e = launch_cfg<256, 256, 2, 4, 4, 32>(p, d->splitk, epi, st); break;
    case 46: e = launch_cfg<256, 128, 4, 2, 2, 64, true>(p, d->splitk, epi, st); break;
    case 47: e = launch_cfg<128, 256, 2, 4, 2, 64, true>(p, d->splitk, epi, st); break;
    case 48: e = launch_cfg<256, 256, 2, 4, 2, 64, true>(p, d->splitk, epi, st); break;
    case 56: e = launch_cfg<256, 128, 4, 2, 4, 32, true>(p, d->splitk, epi, st); break;
    case 57: e = launch_cfg<128, 256, 2, 4, 4, 32, true>(p, d->splitk, epi, st); break;
    case 58: e = launch_cfg<256, 256, 2, 4, 4, 32, true>(p, d->splitk, epi, st); break;
    case 61: e = launch_cfg<128, 128, 2, 2, 2, 32, false>(p, d->splitk, epi, st); break;
    case 64: e = launch_cfg<64, 128, 2, 2, 2, 32, false>(p, d->splitk, epi, st); break;
    case 66: e = launch_cfg<256, 128, 4, 2, 2, 32, true>(p, d->splitk, epi, st); break;
    case 67: e = launch_cfg<128, 256, 2, 4, 2, 32, true>(p, d->splitk, epi, st); break;
    default: diga_set_error("diga_conv_igemm_bf16: tile id %d has no kernel for this entry point", t->id); return DIGA_EINVAL;
  }
#undef DIGA_TILE_CASES
  if (e) return e;
  DIGA_CHECK_LAUNCH("diga_conv_igemm_bf16");
  return DIGA_OK;
}
}  // namespace

extern "C" int diga_conv_igemm_bf16(const diga_conv_desc* d, const void* in, const void* wgt, void* out,
                                    float* stat_slab, const float* stat_shift, void* stream) {
  if (int e = validate(d, "diga_conv_igemm_bf16")) return e;
  DIGA_CHECK_ARG(in && wgt, "diga_conv_igemm_bf16: NULL tensor");
  DIGA_CHECK_ARG((stat_slab == nullptr) == (stat_shift == nullptr), "diga_conv_igemm_bf16: slab and shift go together");
  DIGA_CHECK_ARG(out || stat_slab, "diga_conv_igemm_bf16: no output and no statistics requested");
  DIGA_CHECK_ARG(!(d->out_f32 && stat_slab) || (d->splitk == 1 && out), "diga_conv_igemm_bf16: statistics with fp32 output need splitk == 1 and an output");
  const int epi = d->out_f32 ? (stat_slab ? EPI_F32_STATS : EPI_F32) : (stat_slab ? (out ? EPI_BF16_STATS : EPI_STATS_ONLY) : EPI_BF16);
  return run_conv(d, in, wgt, out, stat_slab, stat_shift, epi, nullptr, nullptr, nullptr, 0, 0, stream);
}

extern "C" int diga_conv_igemm_bf16_apply(const diga_conv_desc* d, const void* in, const void* wgt, void* out, const float* scale,
                                          const float* shift, const void* res, int relu, int nsamples_affine, void* stream) {
  if (int e = validate(d, "diga_conv_igemm_bf16_apply")) return e;
  DIGA_CHECK_ARG(in && wgt && out && scale && shift && res, "diga_conv_igemm_bf16_apply: NULL tensor");
  DIGA_CHECK_ARG(!d->out_f32 && d->splitk == 1 && d->nplanes <= 1, "diga_conv_igemm_bf16_apply: bf16 output, no split-K, no operand planes");
  DIGA_CHECK_ARG(nsamples_affine == 0 || nsamples_affine == d->B, "diga_conv_igemm_bf16_apply: nsamples_affine must be 0 or B");
  return run_conv(d, in, wgt, out, nullptr, nullptr, EPI_FUSED_APPLY, scale, shift, res, nsamples_affine ? d->Cout : 0, relu, stream);
}

extern "C" int diga_conv_dam_ticket_words(const diga_conv_desc* d) {
  if (validate(d, "diga_conv_dam_ticket_words") != DIGA_OK) return DIGA_EINVAL;
  return d->B * (d->cout_pad / 32);          // one counter per (sample, N tile); the narrowest N tile is 32 channels
}

extern "C" int diga_conv_igemm_bf16_dam(const diga_conv_desc* d, const void* in, const void* wgt, void* out, float* stat_slab,
                                        const float* running_mean, const float* running_var, const float* weight, const float* bias,
                                        float lam, float eps, float* scale, float* shift, uint32_t* ticket, void* stream) {
  if (int e = validate(d, "diga_conv_igemm_bf16_dam")) return e;
  DIGA_CHECK_ARG(in && wgt && stat_slab && running_mean && running_var && weight && bias && scale && shift && ticket,
                 "diga_conv_igemm_bf16_dam: NULL pointer");
  DIGA_CHECK_ARG(d->splitk == 1 && (d->out_f32 == 0 || out), "diga_conv_igemm_bf16_dam: splitk == 1; fp32 output needs a buffer");
  const int epi = d->out_f32 ? EPI_F32_STATS : (out ? EPI_BF16_STATS : EPI_STATS_ONLY);
  FinArgs fin;
  fin.ticket = ticket, fin.rvar = running_var, fin.weight = weight, fin.bias = bias, fin.scale = scale, fin.shift = shift;
  fin.lam = lam, fin.eps = eps;
  return run_conv(d, in, wgt, out, stat_slab, running_mean, epi, nullptr, nullptr, nullptr, 0, 0, stream, fin);
}

extern "C" int diga_conv_igemm_bf16_fused_in(const diga_conv_desc* d, const void* raw_in, const float* in_scale, const float* in_shift,
                                             const void* in_res, void* act_out, int nsamples_affine, const void* wgt, void* out,
                                             float* stat_slab, const float* stat_shift, void* stream) {
  if (int e = validate(d, "diga_conv_igemm_bf16_fused_in")) return e;
  DIGA_CHECK_ARG(raw_in && in_scale && in_shift && wgt && out && stat_slab && stat_shift, "diga_conv_igemm_bf16_fused_in: NULL pointer");
  DIGA_CHECK_ARG(!d->out_f32 && d->splitk == 1 && d->nplanes <= 1, "diga_conv_igemm_bf16_fused_in: bf16 output, no split-K, no operand planes");
  DIGA_CHECK_ARG((in_res == nullptr) == (act_out == nullptr), "diga_conv_igemm_bf16_fused_in: in_res and act_out go together");
  DIGA_CHECK_ARG(!in_res || (d->ntaps == 1 && d->stride == 1), "diga_conv_igemm_bf16_fused_in: the residual form is for 1x1 stride-1 convs");
  DIGA_CHECK_ARG(nsamples_affine == 0 || nsamples_affine == d->B, "diga_conv_igemm_bf16_fused_in: nsamples_affine must be 0 or B");
  FusedIn fi;
  fi.amode = in_res ? 2 : 1;
  fi.scale = in_scale, fi.shift = in_shift, fi.res = in_res, fi.act_out = act_out, fi.affine_stride = nsamples_affine ? d->Cin : 0;
  if (d->tile == 0) {                 // auto rule of the fused-input kernels (the caller usually autotunes)
    diga_conv_desc dd = *d;
    dd.tile = d->cout_pad % 128 == 0 ? 21 : 23;
    return run_conv(&dd, raw_in, wgt, out, stat_slab, stat_shift, EPI_BF16_STATS, nullptr, nullptr, nullptr, 0, 0, stream, FinArgs(), fi);
  }
  return run_conv(d, raw_in, wgt, out, stat_slab, stat_shift, EPI_BF16_STATS, nullptr, nullptr, nullptr, 0, 0, stream, FinArgs(), fi);
}

#ifdef DIGA_TILE_CLOCK
extern "C" int diga_debug_tile_clock(unsigned long long* host, int nblocks) {
  if (nblocks > 16384) nblocks = 16384;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tile_clock), (size_t)nblocks * 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif
