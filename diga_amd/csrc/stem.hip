// Stem convolution (adaptseg.py:133,178): 7x7, stride 2, pad 3, 3 -> 64 channels.
// Input is the caller's fp32 NCHW frame (BGR, mean-subtracted), output is the raw bf16 NHWC map that
// bn1+ReLU+maxpool consumes, plus the DAM statistics slab (one unit per block, sample-major: [unit][2][64]).
//
// Cin = 3 makes this a 147-deep dot product per output: too thin for the generic 64-channel gather,
// and only 0.3 % of the frame's FLOPs, so it runs on the fp32 VALU from LDS: a block owns a strip of
// ROWS output rows x 64 output columns x 64 channels, with the 37 KB weight matrix and the strip's
// 13 x 133 x 3 input patch in LDS.
#include "common.h"

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
  else
    hipLaunchKernelGGL(stem_kernel<false>, dim3(ntiles), dim3(256), 0, (hipStream_t)stream, x, wgt, out, stat_slab, stat_shift, H, W, Ho,
                       Wo, tiles_w, tiles_h);
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
