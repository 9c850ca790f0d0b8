// Implicit-GEMM convolution on CDNA4 bf16 MFMA (v_mfma_f32_32x32x16_bf16), NHWC activations.
//
// Replaces the nn.Conv2d call sites of the AdaptSegNet trunk/head (adaptseg.py:85,89,93,97,119-121).
//   GEMM view:  D[m][n] = sum_k A[m][k] * Wt[n][k]
//     m = output pixel (n_img, oh, ow)            M = B*Ho*Wo
//     n = output channel                           N = Cout
//     k = (tap, ci)                                K = ntaps*Cin, one 64-wide K step never crosses a tap
//   A is gathered on the fly (im2col never materialised): row m, tap t reads the 128-byte channel
//   slab of input pixel (oh*stride+dy[t], ow*stride+dx[t]) or zeros outside the image.  Out-of-image
//   and out-of-range rows are produced by the buffer-descriptor range check (offset >= num_records
//   returns 0), so the gather is branch-free.
//
// Block = 256 threads = 4 waves (WM x WN), tile BM x BN x 64, LDS double-buffered with register
// staging (global loads for step k+1 are issued before the MFMAs of step k and written to the other
// LDS buffer after them: one barrier per K step).  LDS rows are 128 B (64 bf16); the 16-byte chunk
// index is XOR-swizzled with (row>>1)&7 so that the ds_read_b128 fragment reads of 32 consecutive
// rows are bank-conflict free.
//
// Epilogue: (1) DAM statistics -- per-channel sum / sum of squares of (x - shift[c]) taken from the
// fp32 accumulators (columns live on lanes, rows in registers, so the reduction is 16 adds + one
// cross-half shuffle + a tiny LDS reduce over the M-waves) written to a per-M-tile slab that
// diga_bn_finalize sums in fixed order (deterministic, no atomics);  (2) the raw output is staged
// through LDS and stored as 16-byte NHWC rows.
#include "common.h"

namespace {

struct ConvKParams {
  const void* in;
  const void* wgt;
  void* out;
  float* slab;
  const float* shift;
  int H, W, Cin, Ho, Wo, Cout, stride, ntaps;
  int M, K, cout_pad, split_row, nk_total, nk_per_split, mtiles, ntiles_n;
  unsigned in_bytes, wgt_bytes;
  short dy[DIGA_MAX_TAPS], dx[DIGA_MAX_TAPS];
};

template <int BM, int BN, int WM, int WN, bool OUT_F32, bool STATS>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvKParams p) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 32, NT = WTN / 32;
  constexpr int A_CH = BM / 32, B_CH = BN / 32;      // 16-byte chunks per thread per K step
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  static_assert(WM * WN == 4 && MT >= 1 && NT >= 1, "bad tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sA = smem;                    // 2 x A_BYTES
  unsigned char* const sB = smem + 2 * A_BYTES;      // 2 x B_BYTES

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, hh = lane >> 5;

  // XCD-aware tile order: blocks are dealt round-robin over the 8 XCDs, so give each XCD a contiguous
  // run of logical tile ids (N tiles of one M tile share the gathered A rows in that XCD's L2).
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mtile = bid / p.ntiles_n, ntile = bid - mtile * p.ntiles_n;
  const int m0 = mtile * BM, n0 = ntile * BN;
  const int ks_begin = blockIdx.y * p.nk_per_split;
  const int ks_end = min(ks_begin + p.nk_per_split, p.nk_total);

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, p.wgt_bytes, 0x00020000);

  // ---- per-thread gather state: rows are fixed for the whole K loop
  const int kc = tid & 7, r0 = tid >> 3;
  int pixoff[A_CH], ih0[A_CH], iw0[A_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int m = m0 + r0 + 32 * i;
    if (m < p.M) {
      const int hw = p.Ho * p.Wo;
      const int n = m / hw, rem = m - n * hw;
      const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
      ih0[i] = oh * p.stride;
      iw0[i] = ow * p.stride;
      pixoff[i] = ((n * p.H + ih0[i]) * p.W + iw0[i]) * p.Cin;
    } else {
      ih0[i] = -100000;  // every tap lands outside the image -> zeros
      iw0[i] = 0;
      pixoff[i] = 0;
    }
  }
  int wrow_off[B_CH];
#pragma unroll
  for (int j = 0; j < B_CH; ++j) wrow_off[j] = (n0 + r0 + 32 * j) * p.K + kc * 8;

  u32x4 ra[A_CH], rb[B_CH];
  auto load_tiles = [&](int ks) {
    const int kk = ks * 64;
    const int tap = kk / p.Cin, c0 = kk - tap * p.Cin;
    const int dy = p.dy[tap], dx = p.dx[tap];
    const int tapoff = (dy * p.W + dx) * p.Cin + c0 + kc * 8;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const bool ok = (unsigned)(ih0[i] + dy) < (unsigned)p.H && (unsigned)(iw0[i] + dx) < (unsigned)p.W;
      const unsigned voff = ok ? (unsigned)(pixoff[i] + tapoff) * 2u : 0x7ffffff0u;
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voff, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_CH; ++j)
      rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)(wrow_off[j] + kk) * 2u, 0, 0);
  };
  const int st_swz = (r0 >> 1) & 7;                  // (row>>1)&7 is the same for row and row+32*i
  const int st_off = r0 * 128 + ((kc ^ st_swz) << 4);
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) *reinterpret_cast<u32x4*>(sA + buf * A_BYTES + st_off + i * 4096) = ra[i];
#pragma unroll
    for (int j = 0; j < B_CH; ++j) *reinterpret_cast<u32x4*>(sB + buf * B_BYTES + st_off + j * 4096) = rb[j];
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int fr_swz = (l31 >> 1) & 7;
  const int fa_row = (wm * WTM + l31) * 128, fb_row = (wn * WTN + l31) * 128;

  if (ks_begin < ks_end) {
    load_tiles(ks_begin);
    store_tiles(0);
  }
  __syncthreads();
  int cur = 0;
  for (int ks = ks_begin; ks < ks_end; ++ks) {
    const bool more = ks + 1 < ks_end;
    if (more) load_tiles(ks + 1);
    const unsigned char* bA = sA + cur * A_BYTES + fa_row;
    const unsigned char* bB = sB + cur * B_BYTES + fb_row;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int coff = ((2 * s + hh) ^ fr_swz) << 4;
      bf16x8 fa[MT], fb[NT];
#pragma unroll
      for (int a = 0; a < MT; ++a) fa[a] = *reinterpret_cast<const bf16x8*>(bA + a * 4096 + coff);
#pragma unroll
      for (int b = 0; b < NT; ++b) fb[b] = *reinterpret_cast<const bf16x8*>(bB + b * 4096 + coff);
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    if (more) store_tiles(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ------------------------------------------------------------------ epilogue
  // accumulator element r of tile (a,b): row = wm*WTM + a*32 + (r&3) + 8*(r>>2) + 4*hh, col = wn*WTN + b*32 + l31
  if constexpr (STATS) {
    float* red = reinterpret_cast<float*>(smem);     // [WM][2 grp][2 stat][BN]
    const bool one_group = (m0 + BM <= p.split_row) || (m0 >= p.split_row && m0 + BM <= p.M);
    const int g_fast = m0 >= p.split_row ? 1 : 0;
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      const int c = n0 + wn * WTN + b * 32 + l31;
      const float kshift = c < p.Cout ? p.shift[c] : 0.f;
      float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
      if (one_group) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float d = acc[a][b][r] - kshift;
            t1 += d;
            t2 = fmaf(d, d, t2);
          }
        s1[g_fast] = t1;
        s2[g_fast] = t2;
      } else {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * WTM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float d = acc[a][b][r] - kshift;
            if (m < p.split_row) {
              s1[0] += d;
              s2[0] = fmaf(d, d, s2[0]);
            } else if (m < p.M) {
              s1[1] += d;
              s2[1] = fmaf(d, d, s2[1]);
            }
          }
      }
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        s1[g] += __shfl_xor(s1[g], 32);
        s2[g] += __shfl_xor(s2[g], 32);
      }
      if (hh == 0) {
        const int col = wn * WTN + b * 32 + l31;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          red[((wm * 2 + g) * 2 + 0) * BN + col] = s1[g];
          red[((wm * 2 + g) * 2 + 1) * BN + col] = s2[g];
        }
      }
    }
    __syncthreads();
    for (int e = tid; e < 4 * BN; e += 256) {
      const int gs = e / BN, col = e - gs * BN;       // gs = g*2 + stat
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) v += red[(w * 4 + gs) * BN + col];
      const int c = n0 + col;
      if (c < p.Cout) p.slab[((size_t)mtile * 4 + gs) * p.Cout + c] = v;
    }
    __syncthreads();
  }

  if constexpr (OUT_F32) {
    float* out = reinterpret_cast<float*>(p.out) + (size_t)blockIdx.y * p.M * p.cout_pad;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const int c = n0 + wn * WTN + b * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * WTM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (m < p.M) out[(size_t)m * p.cout_pad + c] = acc[a][b][r];
        }
      }
  } else {
    unsigned short* sC = reinterpret_cast<unsigned short*>(smem);   // [BM][BN] bf16
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const int col = wn * WTN + b * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * WTM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          sC[row * BN + col] = f32_to_bf16_bits(acc[a][b][r]);
        }
      }
    __syncthreads();
    unsigned short* out = reinterpret_cast<unsigned short*>(p.out);
    constexpr int CPR = BN / 8;                       // 16-byte chunks per tile row
#pragma unroll
    for (int i = 0; i < BM * CPR / 256; ++i) {
      const int id = tid + 256 * i;
      const int row = id / CPR, cc = id - row * CPR;
      const int m = m0 + row, c = n0 + cc * 8;
      if (m < p.M && c < p.Cout)
        *reinterpret_cast<u32x4*>(out + (size_t)m * p.Cout + c) = *reinterpret_cast<const u32x4*>(sC + row * BN + cc * 8);
    }
  }
}

struct TileCfg {
  int id, BM, BN;
};
// id 1: 128x128, 2: 128x64, 3: 64x64, 4: 64x128, 5: 128x32 (head)
const TileCfg kTiles[] = {{1, 128, 128}, {2, 128, 64}, {3, 64, 64}, {4, 64, 128}, {5, 128, 32}};

const TileCfg* find_tile(int id) {
  for (const auto& t : kTiles)
    if (t.id == id) return &t;
  return nullptr;
}

int auto_tile(const diga_conv_desc* d) {
  const int64_t M = (int64_t)d->B * d->Ho * d->Wo;
  if (d->cout_pad % 64 != 0) return 5;
  auto blocks = [&](int bm, int bn) { return (int64_t)cdiv(M, bm) * cdiv(d->cout_pad, bn) * d->splitk; };
  if (d->cout_pad % 128 == 0 && blocks(128, 128) >= 440) return 1;
  if (blocks(128, 64) >= 220) return 2;
  return 3;
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const ConvKParams& p, int splitk, bool out_f32, bool stats, hipStream_t st) {
  constexpr int main_bytes = 2 * (BM + BN) * 128;
  constexpr int epi_bytes = WM * 4 * BN * 4 > BM * BN * 2 ? WM * 4 * BN * 4 : BM * BN * 2;
  constexpr int smem = main_bytes > epi_bytes ? main_bytes : epi_bytes;
  dim3 grid(p.mtiles * p.ntiles_n, splitk), block(256);
  if (out_f32 && !stats)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, true, false>), grid, block, smem, st, p);
  else if (!out_f32 && stats)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, false, true>), grid, block, smem, st, p);
  else if (!out_f32 && !stats)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, false, false>), grid, block, smem, st, p);
  else {
    diga_set_error("diga_conv_igemm_bf16: statistics with fp32 split-K output are not supported");
    return DIGA_EINVAL;
  }
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
  const int64_t in_bytes = (int64_t)d->B * d->H * d->W * d->Cin * 2;
  const int64_t w_bytes = (int64_t)d->cout_pad * d->ntaps * d->Cin * 2;
  DIGA_CHECK_ARG(in_bytes < 0x7ffffff0ll && w_bytes < 0x7ffffff0ll, "%s: tensor exceeds the 2 GiB buffer range", who);
  const int64_t M = (int64_t)d->B * d->Ho * d->Wo;
  DIGA_CHECK_ARG(M * d->cout_pad < 0x7fffffffll, "%s: output too large", who);
  DIGA_CHECK_ARG(d->split_row >= 0 && d->split_row <= M, "%s: split_row out of range", who);
  return DIGA_OK;
}

}  // namespace

extern "C" int diga_conv_num_mtiles(const diga_conv_desc* d) {
  if (validate(d, "diga_conv_num_mtiles") != DIGA_OK) return DIGA_EINVAL;
  const TileCfg* t = find_tile(d->tile ? d->tile : auto_tile(d));
  DIGA_CHECK_ARG(t != nullptr, "diga_conv_num_mtiles: unknown tile id %d", d->tile);
  return cdiv((int64_t)d->B * d->Ho * d->Wo, t->BM);
}

extern "C" int diga_conv_igemm_bf16(const diga_conv_desc* d, const void* in, const void* wgt, void* out,
                                    float* stat_slab, const float* stat_shift, void* stream) {
  if (int e = validate(d, "diga_conv_igemm_bf16")) return e;
  DIGA_CHECK_ARG(in && wgt && out, "diga_conv_igemm_bf16: NULL tensor");
  DIGA_CHECK_ARG((stat_slab == nullptr) == (stat_shift == nullptr), "diga_conv_igemm_bf16: slab and shift go together");
  const TileCfg* t = find_tile(d->tile ? d->tile : auto_tile(d));
  DIGA_CHECK_ARG(t != nullptr, "diga_conv_igemm_bf16: unknown tile id %d", d->tile);
  DIGA_CHECK_ARG(d->cout_pad % t->BN == 0, "diga_conv_igemm_bf16: cout_pad %d not a multiple of tile N %d", d->cout_pad, t->BN);
  ConvKParams p;
  p.in = in;
  p.wgt = wgt;
  p.out = out;
  p.slab = stat_slab;
  p.shift = stat_shift;
  p.H = d->H, p.W = d->W, p.Cin = d->Cin, p.Ho = d->Ho, p.Wo = d->Wo, p.Cout = d->Cout;
  p.stride = d->stride, p.ntaps = d->ntaps;
  p.M = d->B * d->Ho * d->Wo;
  p.K = d->ntaps * d->Cin;
  p.cout_pad = d->cout_pad;
  p.split_row = d->split_row;
  p.nk_total = p.K / 64;
  p.nk_per_split = cdiv(p.nk_total, d->splitk);
  p.mtiles = cdiv(p.M, t->BM);
  p.ntiles_n = d->cout_pad / t->BN;
  p.in_bytes = (unsigned)((int64_t)d->B * d->H * d->W * d->Cin * 2);
  p.wgt_bytes = (unsigned)((int64_t)d->cout_pad * p.K * 2);
  for (int i = 0; i < DIGA_MAX_TAPS; ++i) {
    p.dy[i] = i < d->ntaps ? d->dy[i] : 0;
    p.dx[i] = i < d->ntaps ? d->dx[i] : 0;
  }
  hipStream_t st = (hipStream_t)stream;
  const bool f32 = d->out_f32 != 0, stats = stat_slab != nullptr;
  int e = DIGA_OK;
  switch (t->id) {
    case 1: e = launch_cfg<128, 128, 2, 2>(p, d->splitk, f32, stats, st); break;
    case 2: e = launch_cfg<128, 64, 2, 2>(p, d->splitk, f32, stats, st); break;
    case 3: e = launch_cfg<64, 64, 2, 2>(p, d->splitk, f32, stats, st); break;
    case 4: e = launch_cfg<64, 128, 2, 2>(p, d->splitk, f32, stats, st); break;
    case 5: e = launch_cfg<128, 32, 4, 1>(p, d->splitk, f32, stats, st); break;
  }
  if (e) return e;
  DIGA_CHECK_LAUNCH("diga_conv_igemm_bf16");
  return DIGA_OK;
}
