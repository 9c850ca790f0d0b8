// Stem convolution (adaptseg.py:133,178): 7x7, stride 2, pad 3, 3 -> 64 channels.
// Input is the caller's fp32 NCHW frame (BGR, mean-subtracted), output is the raw bf16 NHWC map that
// bn1+ReLU+maxpool consumes, plus the DAM statistics slab (one unit per block, sample-major: [unit][2][64]).
//
// Cin = 3 makes this a 147-deep dot product per output: too thin for the generic 64-channel gather.  A block owns
// strips of ROWS output rows x 64 output columns x 64 channels (one statistics unit each), the strip's 13 x 133 x 3
// input patch in LDS.  Two kernels:
//  * stem_mfma_kernel (the bf16 product path): fp32 inputs and weights split into bf16 hi + lo, three MFMA terms
//    (hi.hi + hi.lo + lo.hi: 2^-17 relative per product, well inside the bf16 rounding of the output), weight
//    fragments resident in registers, persistent blocks;
//  * stem_kernel<OUT_F32> (the fp32-exact verification mode): fp32 VALU FMAs in the reference's (c, kh, kw) order.
#include "common.h"
#include <type_traits>

namespace {

constexpr int ROWS = 4;        // output rows per block, all four accumulated in registers
constexpr int PR = 2 * (ROWS - 1) + 7;   // input rows the strip needs (13)
constexpr int PW = 134;        // patch row pitch (133 used)

// A thread owns output column `lane` of the strip's ROWS rows x 16 channels (64 accumulators): every weight vector
// read from LDS (4 x b128, broadcast) feeds ROWS x 16 FMAs, every input value 16.  The accumulation order per output
// is (c, kh, kw), the order of the reference's 147-deep dot product layout [c][kh][kw].
template <bool OUT_F32>   // OUT_F32: the verification mode keeps the raw output in fp32 (same arithmetic, no bf16 rounding)
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                   void* __restrict__ out_v, float* __restrict__ slab,
                                                   const float* __restrict__ shift, int H, int W, int Ho, int Wo,
                                                   int tiles_w, int tiles_h) {
  __shared__ __attribute__((aligned(16))) float sw[147 * 64];   // [k][cout]
  __shared__ float sp[3 * PR * PW];                             // [c][input row][col]
  __shared__ float sred[4][2][16];
  const int tid = threadIdx.x, lane = tid & 63, cg = tid >> 6;
  int b = blockIdx.x;
  const int tw = b % tiles_w;
  b /= tiles_w;
  const int th = b % tiles_h, n = b / tiles_h;
  const int ow0 = tw * 64, oh0 = th * ROWS;

  for (int e = tid; e < 147 * 64; e += 256) {
    const int co = e & 63, k = e >> 6;
    sw[e] = wgt[co * 147 + k];
  }
  const float* xin = x + (size_t)n * 3 * H * W;
  for (int e = tid; e < 3 * PR * 133; e += 256) {
    const int j = e % 133, cr = e / 133;
    const int r = cr % PR, c = cr / PR;
    const int ih = oh0 * 2 - 3 + r, iw = ow0 * 2 - 3 + j;
    float v = 0.f;
    if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = xin[((size_t)c * H + ih) * W + iw];
    sp[cr * PW + j] = v;
  }
  __syncthreads();

  float acc[ROWS][16];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
  for (int c = 0; c < 3; ++c)
    for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 7; ++kw) {
        const f32x4* wv = reinterpret_cast<const f32x4*>(&sw[((c * 7 + kh) * 7 + kw) * 64 + cg * 16]);
        const f32x4 w0 = wv[0], w1 = wv[1], w2 = wv[2], w3 = wv[3];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          const float xv = sp[(c * PR + 2 * r + kh) * PW + 2 * lane + kw];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[r][i] = fmaf(xv, w0[i], acc[r][i]);
            acc[r][4 + i] = fmaf(xv, w1[i], acc[r][4 + i]);
            acc[r][8 + i] = fmaf(xv, w2[i], acc[r][8 + i]);
            acc[r][12 + i] = fmaf(xv, w3[i], acc[r][12 + i]);
          }
        }
      }
    }

  float kshift[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) kshift[i] = shift[cg * 16 + i];
  float s1[16], s2[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) s1[i] = s2[i] = 0.f;
  const int ow = ow0 + lane;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int oh = oh0 + r;
    if (oh < Ho && ow < Wo) {
      const size_t e0 = (((size_t)n * Ho + oh) * Wo + ow) * 64 + cg * 16;
      if constexpr (OUT_F32) {
        f32x4* o = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out_v) + e0);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = f32x4{acc[r][4 * i], acc[r][4 * i + 1], acc[r][4 * i + 2], acc[r][4 * i + 3]};
      } else {
        u32x4 lo, hi;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          lo[i] = pack_bf16x2(acc[r][2 * i], acc[r][2 * i + 1]);
          hi[i] = pack_bf16x2(acc[r][8 + 2 * i], acc[r][8 + 2 * i + 1]);
        }
        u32x4* o = reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(out_v) + e0);
        o[0] = lo;
        o[1] = hi;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float d = acc[r][i] - kshift[i];
        s1[i] += d;
        s2[i] = fmaf(d, d, s2[i]);
      }
    }
  }
  // wave reduce over the 64 pixels of the lane dimension (fixed butterfly order -> deterministic)
#pragma unroll
  for (int i = 0; i < 16; ++i) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      s1[i] += __shfl_xor(s1[i], o);
      s2[i] += __shfl_xor(s2[i], o);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      sred[cg][0][i] = s1[i];
      sred[cg][1][i] = s2[i];
    }
  }
  __syncthreads();
  if (tid < 128) {
    // statistics unit = this block: slab[unit][stat][64]; units are numbered sample-major
    const int stat = tid >> 6, c = tid & 63;
    slab[((size_t)blockIdx.x * 2 + stat) * 64 + c] = sred[c >> 4][stat][c & 15];
  }
}

// ---- MFMA form.  GEMM view: out^T[channel][pixel] = W[channel][k] . X[k][pixel] with k = ((c, kh) pair p, kw slot 0..7): the 8
// k of one lane's fragment are the 8 CONSECUTIVE input columns 2 ow - 3 ... 2 ow + 4 of input row 2 oh - 3 + kh, channel c
// (slot 7 meets a zero weight), so a pixel fragment is 16 contiguous bytes of the bf16 patch; 21 pairs -> 6 K steps of 4
// pairs (pairs 21..23 repeat pair 20 against zero weights).  Weights are the A operand, so that a lane's accumulators are
// consecutive CHANNELS of one pixel: MFMA row m of channel tile T is channel 32 half + 8 (m >> 2) + 4 T + (m & 3), and
// acc[0][0..3], acc[1][0..3] of lane (g, px) are channels 32 half + 8 g ... + 7 of pixel px: one 16-byte NHWC store.
// Wave w: channel half w & 1 (weights: 2 tiles x 6 steps x (hi, lo) fragments = 96 registers, loaded once per block),
// rows 2 (w >> 1), 2 (w >> 1) + 1 of the strip = 8 pixel tiles of 16.
// sum over the 16 lanes of a DPP row, in every lane of the row, one fixed order: quad neighbours, quad pairs, the mirrored
// half row, the mirrored row (v_add_f32 with a DPP operand each: no LDS permutes)
__device__ __forceinline__ float row_sum16(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1, 0, 3, 2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2, 3, 0, 1]
  v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror
  return v;
}

constexpr int MPW = 160;       // bf16 patch row pitch: 80 words, consecutive rows 16 banks apart (134 used)

__global__ __launch_bounds__(256) void stem_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                        unsigned short* __restrict__ out, float* __restrict__ slab,
                                                        const float* __restrict__ shift, int H, int W, int Ho, int Wo,
                                                        int tiles_w, int tiles_h, int ntiles) {
  __shared__ __attribute__((aligned(16))) unsigned short sph[3 * PR * MPW];   // hi plane [c][input row][col]
  __shared__ __attribute__((aligned(16))) unsigned short spl[3 * PR * MPW];   // lo plane
  __shared__ float sred[2][2][2][32];                                         // [channel half][row half][stat][channel]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chalf = wave & 1, phalf = wave >> 1, g = lane >> 4, px = lane & 15;

  // weight fragments: lane (m = lane & 15, k block g) holds, for step ks, the 7 taps of pair p = 4 ks + g of channel(m, T)
  bf16x8 wh[2][6], wl[2][6];
#pragma unroll
  for (int T = 0; T < 2; ++T) {
    const int co = chalf * 32 + 8 * (px >> 2) + 4 * T + (px & 3);
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
      const int p = ks * 4 + g;
      float w[8];
#pragma unroll
      for (int kw = 0; kw < 8; ++kw) w[kw] = (p < 21 && kw < 7) ? wgt[co * 147 + p * 7 + kw] : 0.f;
      u32x4 h, l;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        h[i] = pack_bf16x2(w[2 * i], w[2 * i + 1]);
        l[i] = pack_bf16x2(w[2 * i] - __uint_as_float(h[i] << 16), w[2 * i + 1] - __uint_as_float(h[i] & 0xffff0000u));
      }
      wh[T][ks] = __builtin_bit_cast(bf16x8, h);
      wl[T][ks] = __builtin_bit_cast(bf16x8, l);
    }
  }
  int koff[6];                   // element offset of the lane's (c, kh) row per step
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) {
    const int p = min(ks * 4 + g, 20);
    koff[ks] = ((p / 7) * PR + (p % 7)) * MPW;
  }
  float kshift[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) kshift[i] = shift[chalf * 32 + 8 * g + i];

  // patch: a thread owns a PAIR of adjacent columns (one packed conversion, one 4-byte LDS store per plane); a wave owns a
  // row per pass (its validity is wave-uniform): 10 passes x 4 rows for columns 0 ... 127, one more for columns 128 ... 133.
  // The loads of the NEXT strip's patch are issued before this strip's MFMA phase and converted / stored after it.
  constexpr int NPASS = (3 * PR + 3) / 4;
  f32x2 pv[NPASS + 1];
  auto patch_load = [&](int strip) {
    int b = strip;
    const int tw = b % tiles_w;
    b /= tiles_w;
    const int th = b % tiles_h, n = b / tiles_h;
    const int ow0 = tw * 64, oh0 = th * ROWS;
    const float* xin = x + (size_t)n * 3 * H * W;
    const int iwp = ow0 * 2 - 3 + 2 * lane;                     // input column of the pair's first element
    const bool okc0 = (unsigned)iwp < (unsigned)W, okc1 = (unsigned)(iwp + 1) < (unsigned)W;
#pragma unroll
    for (int it = 0; it < NPASS; ++it) {
      const int cr = it * 4 + wave;                             // wave-uniform
      const int r = cr % PR, c = cr / PR, ih = oh0 * 2 - 3 + r;
      pv[it] = f32x2{0.f, 0.f};
      if (cr < 3 * PR && (unsigned)ih < (unsigned)H) {
        const float* row = xin + ((size_t)c * H + ih) * W;
        if (okc0) pv[it][0] = row[iwp];
        if (okc1) pv[it][1] = row[iwp + 1];
      }
    }
    const int cr = tid / 3, jp = 64 + tid % 3;                  // columns 128 ... 133 as 3 pairs x 39 rows (117 threads)
    const int r = cr % PR, c = cr / PR, ih = oh0 * 2 - 3 + r, iw = ow0 * 2 - 3 + 2 * jp;
    pv[NPASS] = f32x2{0.f, 0.f};
    if (tid < 3 * PR * 3 && (unsigned)ih < (unsigned)H) {
      const float* row = xin + ((size_t)c * H + ih) * W;
      if ((unsigned)iw < (unsigned)W) pv[NPASS][0] = row[iw];
      if (jp < 66 && (unsigned)(iw + 1) < (unsigned)W) pv[NPASS][1] = row[iw + 1];   // column 133 is the zero slot's
    }
  };
  auto put = [&](int cr, int jp, f32x2 v) {
    const unsigned h = pack_bf16x2(v[0], v[1]);
    const unsigned l = pack_bf16x2(v[0] - __uint_as_float(h << 16), v[1] - __uint_as_float(h & 0xffff0000u));
    *reinterpret_cast<unsigned*>(sph + cr * MPW + 2 * jp) = h;
    *reinterpret_cast<unsigned*>(spl + cr * MPW + 2 * jp) = l;
  };

  if ((int)blockIdx.x < ntiles) patch_load(blockIdx.x);
  for (int strip = blockIdx.x; strip < ntiles; strip += gridDim.x) {
    int b = strip;
    const int tw = b % tiles_w;
    b /= tiles_w;
    const int th = b % tiles_h, n = b / tiles_h;
    const int ow0 = tw * 64, oh0 = th * ROWS;
    __syncthreads();             // the previous strip's patch and partial sums have been read
#pragma unroll
    for (int it = 0; it < NPASS; ++it)
      if (it * 4 + wave < 3 * PR) put(it * 4 + wave, lane, pv[it]);
    if (tid < 3 * PR * 3) put(tid / 3, 64 + tid % 3, pv[NPASS]);
    __syncthreads();
    if (strip + (int)gridDim.x < ntiles) patch_load(strip + gridDim.x);

    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
#pragma unroll 2
    for (int mt = 0; mt < 8; ++mt) {
      const int r = phalf * 2 + (mt >> 2), owl = (mt & 3) * 16 + px;
      const int base = 2 * r * MPW + 2 * owl;
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 6; ++ks) {
        const unsigned* ph = reinterpret_cast<const unsigned*>(sph + koff[ks] + base);   // 4-byte aligned: even element offset
        const unsigned* pl = reinterpret_cast<const unsigned*>(spl + koff[ks] + base);
        const bf16x8 bh = __builtin_bit_cast(bf16x8, (u32x4{ph[0], ph[1], ph[2], ph[3]}));
        const bf16x8 bl = __builtin_bit_cast(bf16x8, (u32x4{pl[0], pl[1], pl[2], pl[3]}));
#pragma unroll
        for (int T = 0; T < 2; ++T) {                 // smallest terms first
          acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[T][ks], bl, acc[T], 0, 0, 0);
          acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[T][ks], bh, acc[T], 0, 0, 0);
          acc[T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[T][ks], bh, acc[T], 0, 0, 0);
        }
      }
      const int oh = oh0 + r, ow = ow0 + owl;
      if (oh < Ho && ow < Wo) {
        const float a[8] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[1][0], acc[1][1], acc[1][2], acc[1][3]};
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(a[2 * i], a[2 * i + 1]);
        *reinterpret_cast<u32x4*>(out + (((size_t)n * Ho + oh) * Wo + ow) * 64 + chalf * 32 + 8 * g) = o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float d = a[i] - kshift[i];
          s1[i] += d;
          s2[i] = fmaf(d, d, s2[i]);
        }
      }
    }
    // fold the 16 pixel lanes (fixed butterfly), then the two row halves in a fixed order: deterministic
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s1[i] = row_sum16(s1[i]);
      s2[i] = row_sum16(s2[i]);
    }
    if (px == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sred[chalf][phalf][0][8 * g + i] = s1[i];
        sred[chalf][phalf][1][8 * g + i] = s2[i];
      }
    }
    __syncthreads();
    if (tid < 128) {
      // statistics unit = this strip: slab[unit][stat][64]; units are numbered sample-major
      const int stat = tid >> 6, c = tid & 63;
      slab[((size_t)strip * 2 + stat) * 64 + c] = sred[c >> 5][0][stat][c & 31] + sred[c >> 5][1][stat][c & 31];
    }
  }
}

}  // namespace

namespace {
int stem_launch(const float* x, const float* wgt, void* out, float* stat_slab, const float* stat_shift, int B, int H, int W,
                bool out_f32, void* stream) {
  DIGA_CHECK_ARG(B > 0 && H >= 7 && W >= 7, "diga_stem_conv7x7s2: bad dims %dx%dx%d", B, H, W);
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const int tiles_w = cdiv(Wo, 64), tiles_h = cdiv(Ho, ROWS);
  const int ntiles = B * tiles_h * tiles_w;
  if (x == nullptr) return ntiles;
  DIGA_CHECK_ARG(wgt && out && stat_slab && stat_shift, "diga_stem_conv7x7s2: NULL pointer");
  if (out_f32)
    hipLaunchKernelGGL(stem_kernel<true>, dim3(ntiles), dim3(256), 0, (hipStream_t)stream, x, wgt, out, stat_slab, stat_shift, H, W, Ho,
                       Wo, tiles_w, tiles_h);
  else   // persistent: two blocks per CU (208 registers), each sets its weight fragments up once for all its strips
    hipLaunchKernelGGL(stem_mfma_kernel, dim3(ntiles < 512 ? ntiles : 512), dim3(256), 0, (hipStream_t)stream, x, wgt,
                       static_cast<unsigned short*>(out), stat_slab, stat_shift, H, W, Ho, Wo, tiles_w, tiles_h, ntiles);
  DIGA_CHECK_LAUNCH("diga_stem_conv7x7s2");
  return DIGA_OK;
}
}  // namespace

extern "C" int diga_stem_conv7x7s2(const float* x, const float* wgt, void* out, float* stat_slab, const float* stat_shift,
                                   int B, int H, int W, void* stream) {
  return stem_launch(x, wgt, out, stat_slab, stat_shift, B, H, W, false, stream);
}

extern "C" int diga_stem_conv7x7s2_f32(const float* x, const float* wgt, float* out, float* stat_slab, const float* stat_shift,
                                       int B, int H, int W, void* stream) {
  return stem_launch(x, wgt, out, stat_slab, stat_shift, B, H, W, true, stream);
}
