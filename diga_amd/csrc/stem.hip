// Stem convolution (adaptseg.py:133,178): 7x7, stride 2, pad 3, 3 -> 64 channels.
// Input is the caller's fp32 NCHW frame (BGR, mean-subtracted), output is the raw bf16 NHWC map that
// bn1+ReLU+maxpool consumes, plus the DAM statistics slab (one unit per block, sample-major: [unit][2][64]).
//
// Cin = 3 makes this a 147-deep dot product per output: too thin for the generic 64-channel gather,
// and only 0.3 % of the frame's FLOPs, so it runs on the fp32 VALU from LDS: a block owns a strip of
// ROWS output rows x 64 output columns x 64 channels, keeps the 37 KB weight matrix in LDS for the
// whole strip and re-stages the 7 x 133 x 3 input patch per output row.
#include "common.h"

namespace {

constexpr int ROWS = 4;        // output rows per block
constexpr int PW = 134;        // patch row pitch (133 used)

__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                   unsigned short* __restrict__ out, float* __restrict__ slab,
                                                   const float* __restrict__ shift, int H, int W, int Ho, int Wo,
                                                   int tiles_w, int tiles_h) {
  __shared__ __attribute__((aligned(16))) float sw[147 * 64];   // [k][cout]
  __shared__ float sp[3 * 7 * PW];
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
  float kshift[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) kshift[i] = shift[cg * 16 + i];
  float s1[16], s2[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) s1[i] = s2[i] = 0.f;

  const float* xin = x + (size_t)n * 3 * H * W;
  for (int r = 0; r < ROWS; ++r) {
    const int oh = oh0 + r;
    if (oh >= Ho) break;                                   // uniform over the block
    __syncthreads();
    for (int e = tid; e < 3 * 7 * 133; e += 256) {
      const int j = e % 133, ck = e / 133;
      const int kh = ck % 7, c = ck / 7;
      const int ih = oh * 2 - 3 + kh, iw = ow0 * 2 - 3 + j;
      float v = 0.f;
      if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = xin[((size_t)c * H + ih) * W + iw];
      sp[ck * PW + j] = v;
    }
    __syncthreads();
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int ck = 0; ck < 21; ++ck) {
#pragma unroll
      for (int kw = 0; kw < 7; ++kw) {
        const float xv = sp[ck * PW + 2 * lane + kw];
        const f32x4* wv = reinterpret_cast<const f32x4*>(&sw[(ck * 7 + kw) * 64 + cg * 16]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 w4 = wv[q];
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[q * 4 + i] = fmaf(xv, w4[i], acc[q * 4 + i]);
        }
      }
    }
    const int ow = ow0 + lane;
    if (ow < Wo) {
      u32x4 lo, hi;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        lo[i] = pack_bf16x2(acc[2 * i], acc[2 * i + 1]);
        hi[i] = pack_bf16x2(acc[8 + 2 * i], acc[8 + 2 * i + 1]);
      }
      u32x4* o = reinterpret_cast<u32x4*>(out + (((size_t)n * Ho + oh) * Wo + ow) * 64 + cg * 16);
      o[0] = lo;
      o[1] = hi;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float d = acc[i] - kshift[i];
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

extern "C" int diga_stem_conv7x7s2(const float* x, const float* wgt, void* out, float* stat_slab, const float* stat_shift,
                                   int B, int H, int W, void* stream) {
  DIGA_CHECK_ARG(B > 0 && H >= 7 && W >= 7, "diga_stem_conv7x7s2: bad dims %dx%dx%d", B, H, W);
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const int tiles_w = cdiv(Wo, 64), tiles_h = cdiv(Ho, ROWS);
  const int ntiles = B * tiles_h * tiles_w;
  if (x == nullptr) return ntiles;
  DIGA_CHECK_ARG(wgt && out && stat_slab && stat_shift, "diga_stem_conv7x7s2: NULL pointer");
  hipLaunchKernelGGL(stem_kernel, dim3(ntiles), dim3(256), 0, (hipStream_t)stream, x, wgt, (unsigned short*)out, stat_slab,
                     stat_shift, H, W, Ho, Wo, tiles_w, tiles_h);
  DIGA_CHECK_LAUNCH("diga_stem_conv7x7s2");
  return DIGA_OK;
}
