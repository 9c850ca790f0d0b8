// DAM batch norm (SIFABatchNorm2d.forward eval branch, ttda_module.py:299-345) split for the GPU:
//   conv epilogue / stem  -> per-tile shifted sums (slab)
//   diga_bn_finalize      -> fixed-order slab reduce + lambda-mix with the source statistics -> scale/shift
//   diga_bn_apply         -> y = relu(raw*scale + shift (+ residual))   (HBM/L2-bound elementwise)
//   diga_bn_relu_maxpool  -> the stem's bn1 + ReLU + ceil-mode max pool in one pass
#include "common.h"
#include "dam_affine.h"

namespace {

// Slab = [unit][2 (sum, sumsq)][C]; units are sample-major, `ups` units per sample.
// Block = 64 channels (lanes, coalesced slab reads) x FIN_PH unit phases (waves).  Phase w sums units
// u0+w, u0+w+FIN_PH, ... of a unit range in fp64; the phases are then combined through LDS in fixed order
// -> deterministic, and a sample's result does not depend on which batch it is part of (dam_affine.h).

__global__ __launch_bounds__(64 * FIN_PH) void bn_finalize_kernel(const float* __restrict__ slab, int ups, int nsamples, int a_samples,
                                                                 int per_sample, int C, double n_a, double n_b,
                                                                 const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                                 const float* __restrict__ weight, const float* __restrict__ bias,
                                                                 float lam, float eps, float* __restrict__ scale,
                                                                 float* __restrict__ shift) {
  __shared__ double red[FIN_PH][4][64];
  const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  int a0, a1, b1;
  if (per_sample) {                  // blockIdx.y = sample: its own statistics alone (a batch-1 frame of a stream)
    a0 = blockIdx.y * ups, a1 = a0 + ups, b1 = a1;
    scale += (size_t)blockIdx.y * C;
    shift += (size_t)blockIdx.y * C;
  } else {                           // half_first (ttda:313-320): group A = the first a_samples samples, B = the rest
    a0 = 0, a1 = a_samples * ups, b1 = nsamples * ups;
  }
  double s[4];
  slab_group_sums(slab, a0, a1, b1, C, c, c < C, red, s);
  if (ph != 0 || c >= C) return;
  dam_affine_from_sums(s, c, n_a, n_b, rmean, rvar, weight, bias, lam, eps, scale, shift);
}

__global__ void bn_finalize_sums_kernel(const double* __restrict__ sums, int C, double n_a, double n_b, const float* __restrict__ rmean,
                                        const float* __restrict__ rvar, const float* __restrict__ weight,
                                        const float* __restrict__ bias, float lam, float eps, float* __restrict__ scale,
                                        float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  sums += (size_t)blockIdx.y * 4 * C;          // blockIdx.y = set (one global batch of a stream of batches)
  scale += (size_t)blockIdx.y * C;
  shift += (size_t)blockIdx.y * C;
  const double s[4] = {sums[c], sums[C + c], sums[2 * C + c], sums[3 * C + c]};
  dam_affine_from_sums(s, c, n_a, n_b, rmean, rvar, weight, bias, lam, eps, scale, shift);
}

__global__ __launch_bounds__(64 * FIN_PH) void bn_reduce_slab_kernel(const float* __restrict__ slab, int ups, int nsamples,
                                                                    int a_samples, int per_sample, int C,
                                                                    double* __restrict__ sums) {
  __shared__ double red[FIN_PH][4][64];
  const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  double s[4];
  if (per_sample) {     // blockIdx.y = local sample = its own set: all of it is group A (a_samples > 0) or group B
    const int u0 = blockIdx.y * ups, u1 = u0 + ups;
    slab_group_sums(slab, u0, a_samples > 0 ? u1 : u0, u1, C, c, c < C, red, s);
    sums += (size_t)blockIdx.y * 4 * C;
  } else {
    slab_group_sums(slab, 0, a_samples * ups, nsamples * ups, C, c, c < C, red, s);
  }
  if (ph != 0 || c >= C) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) sums[k * C + c] = s[k];
}

// rows x C elementwise.  A thread owns one 8-channel octet (16 B) and walks rows, so its 8 scale/shift
// pairs stay in registers; consecutive threads take consecutive octets of a row (coalesced 16-B accesses).
// MODE 0: no residual, 1: normalised residual, 2: raw residual with its own affine (downsample branch).
template <int MODE>
__global__ __launch_bounds__(256) void bn_apply_kernel(const u32x4* __restrict__ raw, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const u32x4* __restrict__ res,
                                                       const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                       u32x4* __restrict__ out, int64_t M, int C8, int relu,
                                                       int rows_per_block, int per_sample) {
  const int oct = threadIdx.x % C8, rsub = threadIdx.x / C8, rstep = 256 / C8;
  // blockIdx.y = sample; M = rows of one sample.  per_sample: every sample has its own affine ([nsamples][C])
  const size_t sbase = (size_t)blockIdx.y * M * C8;
  raw += sbase;
  out += sbase;
  if (MODE != 0) res += sbase;
  if (per_sample) {
    scale += (size_t)blockIdx.y * C8 * 8;
    shift += (size_t)blockIdx.y * C8 * 8;
    if (MODE == 2) {
      rscale += (size_t)blockIdx.y * C8 * 8;
      rshift += (size_t)blockIdx.y * C8 * 8;
    }
  }
  float sc[8], sh[8], rsc[8], rsh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sc[i] = scale[oct * 8 + i];
    sh[i] = shift[oct * 8 + i];
    if (MODE == 2) {
      rsc[i] = rscale[oct * 8 + i];
      rsh[i] = rshift[oct * 8 + i];
    }
  }
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t row1 = row0 + rows_per_block < M ? row0 + rows_per_block : M;
  for (int64_t r = row0 + rsub; r < row1; r += rstep) {
    const int64_t idx = r * C8 + oct;
    float v[8], q[8];
    // raw and the residual are read here for the last time: non-temporal loads keep these dead lines from displacing
    // live ones (the block output written below is the next conv's input) -- +4 % whole-step throughput, same-box A/B
    unpack_bf16x8(__builtin_nontemporal_load(&raw[idx]), v);
    if (MODE != 0) unpack_bf16x8(__builtin_nontemporal_load(&res[idx]), q);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float y = fmaf(v[i], sc[i], sh[i]);
      if (MODE == 1) y += q[i];
      if (MODE == 2) y += fmaf(q[i], rsc[i], rsh[i]);
      v[i] = relu ? fmaxf(y, 0.f) : y;
    }
    out[idx] = pack_bf16x8(v);
  }
}

// diga_bn_finalize + diga_bn_apply in ONE launch, for small passes (one frame per step): there both kernels sit at their launch
// floor (5.1 + 7.0 us on a 4-17 MB tensor), and the slab is small enough for every block to reduce it again.  Block = 64 channels
// x FIN_PH phases, as bn_finalize_kernel: it derives the affine of its 64 channels (same sums, same order, same arithmetic:
// bit-identical scale / shift), keeps it in LDS, and applies it to its chunk of rows -- 8 lanes per row (128 contiguous bytes),
// 128 rows per pass.  The blocks of row chunk 0 also publish scale / shift.  MODE 0: no residual, 1: normalised residual.
template <int MODE>
__global__ __launch_bounds__(64 * FIN_PH) void bn_finalize_apply_kernel(
    const float* __restrict__ slab, int ups, int nsamples, int a_samples, int per_sample, int C, double n_a, double n_b,
    const float* __restrict__ rmean, const float* __restrict__ rvar, const float* __restrict__ weight, const float* __restrict__ bias,
    float lam, float eps, float* __restrict__ scale, float* __restrict__ shift, const u32x4* __restrict__ raw,
    const u32x4* __restrict__ res, u32x4* __restrict__ out, int64_t Ms, int relu, int rows_per_block) {
  __shared__ double red[FIN_PH][4][64];
  __shared__ float lsc[64], lsh[64];
  const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int cbase = blockIdx.x * 64, c = cbase + lane;
  int a0, a1, b1;
  if (per_sample) {                  // blockIdx.z = sample
    a0 = blockIdx.z * ups, a1 = a0 + ups, b1 = a1;
    scale += (size_t)blockIdx.z * C;
    shift += (size_t)blockIdx.z * C;
  } else {
    a0 = 0, a1 = a_samples * ups, b1 = nsamples * ups;
  }
  double s[4];
  slab_group_sums(slab, a0, a1, b1, C, c, c < C, red, s);
  if (ph == 0 && c < C) {
    dam_affine_from_sums(s, c, n_a, n_b, rmean, rvar, weight, bias, lam, eps, lsc, lsh, lane);
    if (blockIdx.y == 0) {
      scale[c] = lsc[lane];
      shift[c] = lsh[lane];
    }
  }
  __syncthreads();
  const int oct = threadIdx.x & 7, rsub = threadIdx.x >> 3, C8 = C >> 3;
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sc[i] = lsc[oct * 8 + i];
    sh[i] = lsh[oct * 8 + i];
  }
  const size_t sbase = (size_t)blockIdx.z * Ms * C8 + blockIdx.x * 8 + oct;       // blockIdx.z = 0 unless per_sample
  const int64_t row0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t row1 = row0 + rows_per_block < Ms ? row0 + rows_per_block : Ms;
  for (int64_t r = row0 + rsub; r < row1; r += 128) {
    const size_t idx = sbase + (size_t)r * C8;
    float v[8], q[8];
    unpack_bf16x8(__builtin_nontemporal_load(&raw[idx]), v);
    if (MODE != 0) unpack_bf16x8(__builtin_nontemporal_load(&res[idx]), q);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float y = fmaf(v[i], sc[i], sh[i]);
      if (MODE == 1) y += q[i];
      v[i] = relu ? fmaxf(y, 0.f) : y;
    }
    out[idx] = pack_bf16x8(v);
  }
}

__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const u32x4* __restrict__ raw, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, u32x4* __restrict__ out,
                                                              int B, int H, int W, int C8, int Ho, int Wo, int per_sample) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * Ho * Wo * C8;
  if (idx >= total) return;
  const int oct = (int)(idx % C8);
  int64_t pix = idx / C8;
  const int ow = (int)(pix % Wo);
  pix /= Wo;
  const int oh = (int)(pix % Ho), n = (int)(pix / Ho);
  const int aoff = per_sample ? n * C8 * 8 : 0;
  float sc[8], sh[8], best[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sc[i] = scale[aoff + oct * 8 + i];
    sh[i] = shift[aoff + oct * 8 + i];
    best[i] = -INFINITY;
  }
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int ih = oh * 2 - 1 + kh;
    if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int iw = ow * 2 - 1 + kw;
      if ((unsigned)iw >= (unsigned)W) continue;
      float v[8];
      unpack_bf16x8(raw[((int64_t)(n * H + ih) * W + iw) * C8 + oct], v);
#pragma unroll
      for (int i = 0; i < 8; ++i) best[i] = fmaxf(best[i], fmaxf(fmaf(v[i], sc[i], sh[i]), 0.f));
    }
  }
  out[idx] = pack_bf16x8(best);
}

// [B][HW][C] bf16 -> [B][C][HW] fp32 through a 32x32 LDS tile
__global__ void nhwc_to_nchw_kernel(const unsigned short* __restrict__ in, float* __restrict__ out, int HW, int C) {
  __shared__ float t[32][33];
  const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int p = p0 + i, c = c0 + tx;
    t[i][tx] = (p < HW && c < C) ? bf16_bits_to_f32(in[((size_t)b * HW + p) * C + c]) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, p = p0 + tx;
    if (p < HW && c < C) out[((size_t)b * C + c) * HW + p] = t[tx][i];
  }
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, int HW, int C) {
  __shared__ float t[32][33];
  const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, p = p0 + tx;
    t[i][tx] = (p < HW && c < C) ? in[((size_t)b * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int p = p0 + i, c = c0 + tx;
    if (p < HW && c < C) out[((size_t)b * HW + p) * C + c] = f32_to_bf16_bits(t[tx][i]);
  }
}

// Stand-alone DAM statistics for an fp32 NCHW tensor (the SIFABatchNorm2d module used on its own):
// one block per channel; per-thread shifted sums, 64-lane wavefront shuffle reduction, then a fixed-order
// combine of the 4 waves through LDS.  Writes [2 groups][2][C] = a slab of two one-unit samples for diga_bn_finalize.
__global__ __launch_bounds__(256) void bn_stats_nchw_kernel(const float* __restrict__ x, int B, int C, int HW, int split_b,
                                                            const float* __restrict__ shift, float* __restrict__ slab) {
  __shared__ double red[4][4];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float K = shift[c];
  double acc[4] = {0, 0, 0, 0};
  for (int b = 0; b < B; ++b) {
    const float* p = x + ((size_t)b * C + c) * HW;
    float s1 = 0.f, s2 = 0.f;
    for (int i = tid; i < HW; i += 256) {
      const float d = p[i] - K;
      s1 += d;
      s2 = fmaf(d, d, s2);
    }
    const int g = b < split_b ? 0 : 1;
    acc[g * 2] += (double)s1;
    acc[g * 2 + 1] += (double)s2;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc[k] += __shfl_xor(acc[k], o);
    if (lane == 0) red[wave][k] = acc[k];
  }
  __syncthreads();
  if (tid < 4) slab[(size_t)tid * C + c] = (float)(red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
}

__global__ void bn_apply_nchw_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                     float* __restrict__ out, int C, int HW, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)((i / HW) % C);
  out[i] = fmaf(x[i], scale[c], shift[c]);
}


// ---------------------------------------------------------------------------------------------
// Verification mode (fp32 activations carried as bf16 operand planes, see conv_igemm.hip): the same three elementwise
// stages on fp32 data.  An fp32 value is stored for the next conv as nplanes bf16 planes p0 = bf16(y), p1 = bf16(y - p0),
// p2 = bf16(y - p0 - p1) (the subtractions are exact in fp32), plus an fp32 copy for residuals / SAM.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void split_planes_store(float y, unsigned short* __restrict__ planes, int nplanes, size_t plane_stride,
                                                   size_t idx) {
  float r = y;
  for (int q = 0; q < nplanes; ++q) {
    const unsigned short b = f32_to_bf16_bits(r);
    planes[q * plane_stride + idx] = b;
    r -= bf16_bits_to_f32(b);
  }
}

// MODE as bn_apply_kernel; thread = one element (this path is about exactness, not speed)
template <int MODE>
__global__ __launch_bounds__(256) void bn_apply_f32_kernel(const float* __restrict__ raw, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ res,
                                                           const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                           float* __restrict__ out_f32, unsigned short* __restrict__ planes,
                                                           int nplanes, int64_t total, int C, int relu, int64_t sample_elems) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  // sample_elems > 0: every sample (sample_elems consecutive elements) has its own affine [nsamples][C]
  const int c = (int)(i % C) + (sample_elems > 0 ? (int)(i / sample_elems) * C : 0);
  float y = fmaf(raw[i], scale[c], shift[c]);
  if (MODE == 1) y += res[i];
  if (MODE == 2) y += fmaf(res[i], rscale[c], rshift[c]);
  if (relu) y = fmaxf(y, 0.f);
  if (out_f32) out_f32[i] = y;
  if (planes) split_planes_store(y, planes, nplanes, (size_t)total, (size_t)i);
}

__global__ __launch_bounds__(256) void bn_relu_maxpool_f32_kernel(const float* __restrict__ raw, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, float* __restrict__ out_f32,
                                                                  unsigned short* __restrict__ planes, int nplanes, int B, int H,
                                                                  int W, int C, int Ho, int Wo, int per_sample) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * Ho * Wo * C;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  int64_t pix = idx / C;
  const int ow = (int)(pix % Wo);
  pix /= Wo;
  const int oh = (int)(pix % Ho), n = (int)(pix / Ho);
  const int ac = c + (per_sample ? n * C : 0);
  const float sc = scale[ac], sh = shift[ac];
  float best = -INFINITY;
  for (int kh = 0; kh < 3; ++kh) {
    const int ih = oh * 2 - 1 + kh;
    if ((unsigned)ih >= (unsigned)H) continue;
    for (int kw = 0; kw < 3; ++kw) {
      const int iw = ow * 2 - 1 + kw;
      if ((unsigned)iw >= (unsigned)W) continue;
      best = fmaxf(best, fmaxf(fmaf(raw[((int64_t)(n * H + ih) * W + iw) * C + c], sc, sh), 0.f));
    }
  }
  if (out_f32) out_f32[idx] = best;
  if (planes) split_planes_store(best, planes, nplanes, (size_t)total, (size_t)idx);
}

__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, unsigned short* __restrict__ planes,
                                                           int nplanes, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) split_planes_store(x[i], planes, nplanes, (size_t)n, (size_t)i);
}

// [B][C][HW] fp32 <-> [B][HW][C] fp32 through a 32x32 LDS tile
__global__ void nchw_to_nhwc_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int HW, int C) {
  __shared__ float t[32][33];
  const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, p = p0 + tx;
    t[i][tx] = (p < HW && c < C) ? in[((size_t)b * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int p = p0 + i, c = c0 + tx;
    if (p < HW && c < C) out[((size_t)b * HW + p) * C + c] = t[tx][i];
  }
}

__global__ void nhwc_to_nchw_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int HW, int C) {
  __shared__ float t[32][33];
  const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int p = p0 + i, c = c0 + tx;
    t[i][tx] = (p < HW && c < C) ? in[((size_t)b * HW + p) * C + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, p = p0 + tx;
    if (p < HW && c < C) out[((size_t)b * C + c) * HW + p] = t[tx][i];
  }
}

}  // namespace

extern "C" int diga_bn_finalize(const float* slab, int units_per_sample, int nsamples, int a_samples, int per_sample, int C,
                                double n_a, double n_b, const float* running_mean, const float* running_var,
                                const float* weight, const float* bias, float lam, float eps, float* scale, float* shift,
                                void* stream) {
  DIGA_CHECK_ARG(slab && running_mean && running_var && weight && bias && scale && shift, "diga_bn_finalize: NULL pointer");
  DIGA_CHECK_ARG(units_per_sample > 0 && nsamples > 0 && C > 0 && n_a > 0 && n_b >= 0,
                 "diga_bn_finalize: bad sizes units_per_sample=%d nsamples=%d C=%d", units_per_sample, nsamples, C);
  DIGA_CHECK_ARG(a_samples >= 0 && a_samples <= nsamples, "diga_bn_finalize: a_samples=%d outside [0,%d]", a_samples, nsamples);
  DIGA_CHECK_ARG(!per_sample || n_b == 0, "diga_bn_finalize: per_sample statistics have no group B");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 64), per_sample ? nsamples : 1), dim3(64 * FIN_PH), 0, (hipStream_t)stream,
                     slab, units_per_sample, nsamples, a_samples, per_sample, C, n_a, n_b, running_mean, running_var, weight, bias,
                     lam, eps, scale, shift);
  DIGA_CHECK_LAUNCH("diga_bn_finalize");
  return DIGA_OK;
}

extern "C" int diga_bn_finalize_sums(const double* sums, int nsets, int C, double n_a, double n_b, const float* running_mean,
                                     const float* running_var, const float* weight, const float* bias, float lam, float eps,
                                     float* scale, float* shift, void* stream) {
  DIGA_CHECK_ARG(sums && running_mean && running_var && weight && bias && scale && shift, "diga_bn_finalize_sums: NULL pointer");
  DIGA_CHECK_ARG(nsets > 0 && C > 0 && n_a > 0 && n_b >= 0, "diga_bn_finalize_sums: bad sizes nsets=%d C=%d", nsets, C);
  hipLaunchKernelGGL(bn_finalize_sums_kernel, dim3(cdiv(C, 256), nsets), dim3(256), 0, (hipStream_t)stream, sums, C, n_a, n_b, running_mean,
                     running_var, weight, bias, lam, eps, scale, shift);
  DIGA_CHECK_LAUNCH("diga_bn_finalize_sums");
  return DIGA_OK;
}

extern "C" int diga_bn_reduce_slab(const float* slab, int units_per_sample, int nsamples, int a_samples, int per_sample, int C,
                                   double* sums, void* stream) {
  DIGA_CHECK_ARG(slab && sums && units_per_sample > 0 && nsamples > 0 && C > 0 && a_samples >= 0 && a_samples <= nsamples,
                 "diga_bn_reduce_slab: bad argument");
  hipLaunchKernelGGL(bn_reduce_slab_kernel, dim3(cdiv(C, 64), per_sample ? nsamples : 1), dim3(64 * FIN_PH), 0, (hipStream_t)stream,
                     slab, units_per_sample, nsamples, a_samples, per_sample, C, sums);
  DIGA_CHECK_LAUNCH("diga_bn_reduce_slab");
  return DIGA_OK;
}

extern "C" int diga_bn_apply(const void* raw, const float* scale, const float* shift, const void* res,
                             const float* res_scale, const float* res_shift, void* out, int64_t M, int C, int relu,
                             int nsamples_affine, void* stream) {
  DIGA_CHECK_ARG(raw && scale && shift && out, "diga_bn_apply: NULL pointer");
  DIGA_CHECK_ARG(M > 0 && C >= 64 && C <= 2048 && (C & (C - 1)) == 0, "diga_bn_apply: C=%d must be a power of two in [64,2048]", C);
  DIGA_CHECK_ARG((res_scale == nullptr) == (res_shift == nullptr), "diga_bn_apply: res_scale/res_shift go together");
  DIGA_CHECK_ARG(res != nullptr || res_scale == nullptr, "diga_bn_apply: residual affine without residual");
  DIGA_CHECK_ARG(nsamples_affine >= 0 && (nsamples_affine == 0 || M % nsamples_affine == 0),
                 "diga_bn_apply: M=%lld is not a whole number of %d samples", (long long)M, nsamples_affine);
  const int ns = nsamples_affine > 0 ? nsamples_affine : 1;
  const int64_t Ms = M / ns;                     // rows handled per grid.y slice
  const int C8 = C / 8, rstep = 256 / C8;
  // Rows per thread: 2 (4 for narrow tensors).  Many small blocks in flight beat long per-thread row walks -- same-box sweep at
  // 15 frames (profiles/experiments/bench_apply.py): C = 256: 23.7 -> 21.3 us, C = 1024 + residual: 129.8 -> 113.1 us (6.8 TB/s of
  // algorithmic bytes), C = 2048 + residual: 316 -> 261 us -- but not one row (the thread's 16 scale / shift words are loaded per block)
  // the downsample form loads a second scale / shift set per block: four rows per thread there (619 -> 567 us over its four
  // launches at 15 frames; eight rows: the same)
  int rpb = (C8 <= 16 || res_scale ? 4 : 2) * rstep;
  if (cdiv(Ms, rpb) > 65535) rpb = cdiv(cdiv(Ms, 65535), rstep) * rstep;
  const dim3 grid(cdiv(Ms, rpb), ns), block(256);
  hipStream_t st = (hipStream_t)stream;
  const u32x4* r = (const u32x4*)raw;
  const u32x4* q = (const u32x4*)res;
  const int ps = nsamples_affine > 0;
  if (!res)
    hipLaunchKernelGGL(bn_apply_kernel<0>, grid, block, 0, st, r, scale, shift, q, res_scale, res_shift, (u32x4*)out, Ms, C8, relu, rpb, ps);
  else if (!res_scale)
    hipLaunchKernelGGL(bn_apply_kernel<1>, grid, block, 0, st, r, scale, shift, q, res_scale, res_shift, (u32x4*)out, Ms, C8, relu, rpb, ps);
  else
    hipLaunchKernelGGL(bn_apply_kernel<2>, grid, block, 0, st, r, scale, shift, q, res_scale, res_shift, (u32x4*)out, Ms, C8, relu, rpb, ps);
  DIGA_CHECK_LAUNCH("diga_bn_apply");
  return DIGA_OK;
}

extern "C" int diga_bn_finalize_apply(const float* slab, int units_per_sample, int nsamples, int a_samples, int per_sample, int C,
                                      double n_a, double n_b, const float* running_mean, const float* running_var, const float* weight,
                                      const float* bias, float lam, float eps, float* scale, float* shift, const void* raw,
                                      const void* res, void* out, int64_t M, int relu, void* stream) {
  DIGA_CHECK_ARG(slab && running_mean && running_var && weight && bias && scale && shift && raw && out,
                 "diga_bn_finalize_apply: NULL pointer");
  DIGA_CHECK_ARG(units_per_sample > 0 && nsamples > 0 && n_a > 0 && n_b >= 0 && M > 0,
                 "diga_bn_finalize_apply: bad sizes units_per_sample=%d nsamples=%d M=%lld", units_per_sample, nsamples, (long long)M);
  DIGA_CHECK_ARG(C >= 64 && C <= 2048 && C % 64 == 0, "diga_bn_finalize_apply: C=%d must be a multiple of 64 in [64,2048]", C);
  DIGA_CHECK_ARG(a_samples >= 0 && a_samples <= nsamples, "diga_bn_finalize_apply: a_samples=%d outside [0,%d]", a_samples, nsamples);
  DIGA_CHECK_ARG(!per_sample || (n_b == 0 && M % nsamples == 0), "diga_bn_finalize_apply: per_sample statistics: no group B, M a whole number of samples");
  const int nz = per_sample ? nsamples : 1;
  const int64_t Ms = M / nz;
  // ~192 blocks in all (every block repeats the slab reduction of its 64 channels), whole 128-row passes per block
  int chunks = cdiv(192, (C / 64) * nz);
  if (chunks > cdiv(Ms, 128)) chunks = cdiv(Ms, 128);
  const int rpb = cdiv(cdiv(Ms, chunks), 128) * 128;
  const dim3 grid(C / 64, cdiv(Ms, rpb), nz), block(64 * FIN_PH);
  hipStream_t st = (hipStream_t)stream;
  if (!res)
    hipLaunchKernelGGL(bn_finalize_apply_kernel<0>, grid, block, 0, st, slab, units_per_sample, nsamples, a_samples, per_sample, C, n_a, n_b,
                       running_mean, running_var, weight, bias, lam, eps, scale, shift, (const u32x4*)raw, (const u32x4*)res, (u32x4*)out, Ms,
                       relu, rpb);
  else
    hipLaunchKernelGGL(bn_finalize_apply_kernel<1>, grid, block, 0, st, slab, units_per_sample, nsamples, a_samples, per_sample, C, n_a, n_b,
                       running_mean, running_var, weight, bias, lam, eps, scale, shift, (const u32x4*)raw, (const u32x4*)res, (u32x4*)out, Ms,
                       relu, rpb);
  DIGA_CHECK_LAUNCH("diga_bn_finalize_apply");
  return DIGA_OK;
}

extern "C" int diga_bn_relu_maxpool(const void* raw, const float* scale, const float* shift, void* out, int B, int H, int W,
                                    int C, int Ho, int Wo, int per_sample, void* stream) {
  DIGA_CHECK_ARG(raw && scale && shift && out, "diga_bn_relu_maxpool: NULL pointer");
  DIGA_CHECK_ARG(B > 0 && H > 1 && W > 1 && C % 8 == 0, "diga_bn_relu_maxpool: bad dims");
  // PyTorch ceil_mode rule for k=3, s=2, p=1: o = ceil((H-1)/2)+1, minus one if the last window starts beyond the input
  auto osz = [](int L) {
    int o = (L + 2 - 3 + 1) / 2 + 1;
    if ((o - 1) * 2 >= L + 1) --o;
    return o;
  };
  DIGA_CHECK_ARG(Ho == osz(H) && Wo == osz(W), "diga_bn_relu_maxpool: output %dx%d does not match ceil-mode pooling of %dx%d", Ho, Wo, H, W);
  const int64_t total = (int64_t)B * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(bn_relu_maxpool_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)raw, scale,
                     shift, (u32x4*)out, B, H, W, C / 8, Ho, Wo, per_sample);
  DIGA_CHECK_LAUNCH("diga_bn_relu_maxpool");
  return DIGA_OK;
}

extern "C" int diga_nhwc_bf16_to_nchw_f32(const void* in, float* out, int B, int HW, int C, void* stream) {
  DIGA_CHECK_ARG(in && out && B > 0 && HW > 0 && C > 0, "diga_nhwc_bf16_to_nchw_f32: bad argument");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), B), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)in, out, HW, C);
  DIGA_CHECK_LAUNCH("diga_nhwc_bf16_to_nchw_f32");
  return DIGA_OK;
}

extern "C" int diga_nchw_f32_to_nhwc_bf16(const float* in, void* out, int B, int HW, int C, void* stream) {
  DIGA_CHECK_ARG(in && out && B > 0 && HW > 0 && C > 0, "diga_nchw_f32_to_nhwc_bf16: bad argument");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), B), dim3(256), 0, (hipStream_t)stream, in,
                     (unsigned short*)out, HW, C);
  DIGA_CHECK_LAUNCH("diga_nchw_f32_to_nhwc_bf16");
  return DIGA_OK;
}

extern "C" int diga_bn_stats_nchw_f32(const float* x, int B, int C, int HW, int split_b, const float* stat_shift, float* slab,
                                      void* stream) {
  DIGA_CHECK_ARG(x && stat_shift && slab && B > 0 && C > 0 && HW > 0 && split_b >= 0 && split_b <= B, "diga_bn_stats_nchw_f32: bad argument");
  hipLaunchKernelGGL(bn_stats_nchw_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, B, C, HW, split_b, stat_shift, slab);
  DIGA_CHECK_LAUNCH("diga_bn_stats_nchw_f32");
  return DIGA_OK;
}

extern "C" int diga_bn_apply_nchw_f32(const float* x, const float* scale, const float* shift, float* out, int B, int C, int HW,
                                      void* stream) {
  DIGA_CHECK_ARG(x && scale && shift && out && B > 0 && C > 0 && HW > 0, "diga_bn_apply_nchw_f32: bad argument");
  const int64_t total = (int64_t)B * C * HW;
  hipLaunchKernelGGL(bn_apply_nchw_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, out, C, HW,
                     total);
  DIGA_CHECK_LAUNCH("diga_bn_apply_nchw_f32");
  return DIGA_OK;
}

extern "C" int diga_bn_apply_f32(const float* raw, const float* scale, const float* shift, const float* res, const float* res_scale,
                                 const float* res_shift, float* out_f32, void* out_planes, int nplanes, int64_t M, int C, int relu,
                                 int nsamples_affine, void* stream) {
  DIGA_CHECK_ARG(raw && scale && shift && (out_f32 || out_planes), "diga_bn_apply_f32: NULL pointer");
  DIGA_CHECK_ARG(M > 0 && C > 0 && (!out_planes || (nplanes >= 1 && nplanes <= 3)), "diga_bn_apply_f32: bad sizes / nplanes");
  DIGA_CHECK_ARG((res_scale == nullptr) == (res_shift == nullptr), "diga_bn_apply_f32: res_scale/res_shift go together");
  DIGA_CHECK_ARG(res != nullptr || res_scale == nullptr, "diga_bn_apply_f32: residual affine without residual");
  DIGA_CHECK_ARG(nsamples_affine >= 0 && (nsamples_affine == 0 || M % nsamples_affine == 0),
                 "diga_bn_apply_f32: M=%lld is not a whole number of %d samples", (long long)M, nsamples_affine);
  const int64_t se = nsamples_affine > 0 ? M / nsamples_affine * C : 0;
  const int64_t total = M * C;
  const dim3 grid(cdiv(total, 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  unsigned short* pl = (unsigned short*)out_planes;
  if (!res)
    hipLaunchKernelGGL(bn_apply_f32_kernel<0>, grid, block, 0, st, raw, scale, shift, res, res_scale, res_shift, out_f32, pl, nplanes, total, C, relu, se);
  else if (!res_scale)
    hipLaunchKernelGGL(bn_apply_f32_kernel<1>, grid, block, 0, st, raw, scale, shift, res, res_scale, res_shift, out_f32, pl, nplanes, total, C, relu, se);
  else
    hipLaunchKernelGGL(bn_apply_f32_kernel<2>, grid, block, 0, st, raw, scale, shift, res, res_scale, res_shift, out_f32, pl, nplanes, total, C, relu, se);
  DIGA_CHECK_LAUNCH("diga_bn_apply_f32");
  return DIGA_OK;
}

extern "C" int diga_bn_relu_maxpool_f32(const float* raw, const float* scale, const float* shift, float* out_f32, void* out_planes,
                                        int nplanes, int B, int H, int W, int C, int Ho, int Wo, int per_sample, void* stream) {
  DIGA_CHECK_ARG(raw && scale && shift && (out_f32 || out_planes), "diga_bn_relu_maxpool_f32: NULL pointer");
  DIGA_CHECK_ARG(B > 0 && H > 1 && W > 1 && C > 0 && (!out_planes || (nplanes >= 1 && nplanes <= 3)), "diga_bn_relu_maxpool_f32: bad dims");
  auto osz = [](int L) {
    int o = (L + 2 - 3 + 1) / 2 + 1;
    if ((o - 1) * 2 >= L + 1) --o;
    return o;
  };
  DIGA_CHECK_ARG(Ho == osz(H) && Wo == osz(W), "diga_bn_relu_maxpool_f32: output %dx%d does not match ceil-mode pooling of %dx%d", Ho, Wo, H, W);
  const int64_t total = (int64_t)B * Ho * Wo * C;
  hipLaunchKernelGGL(bn_relu_maxpool_f32_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, raw, scale, shift, out_f32,
                     (unsigned short*)out_planes, nplanes, B, H, W, C, Ho, Wo, per_sample);
  DIGA_CHECK_LAUNCH("diga_bn_relu_maxpool_f32");
  return DIGA_OK;
}

extern "C" int diga_split_planes_f32(const float* x, void* planes, int nplanes, int64_t n, void* stream) {
  DIGA_CHECK_ARG(x && planes && n > 0 && nplanes >= 1 && nplanes <= 3, "diga_split_planes_f32: bad argument");
  hipLaunchKernelGGL(split_planes_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)planes, nplanes, n);
  DIGA_CHECK_LAUNCH("diga_split_planes_f32");
  return DIGA_OK;
}

extern "C" int diga_nchw_f32_to_nhwc_f32(const float* in, float* out, int B, int HW, int C, void* stream) {
  DIGA_CHECK_ARG(in && out && B > 0 && HW > 0 && C > 0, "diga_nchw_f32_to_nhwc_f32: bad argument");
  hipLaunchKernelGGL(nchw_to_nhwc_f32_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), B), dim3(256), 0, (hipStream_t)stream, in, out, HW, C);
  DIGA_CHECK_LAUNCH("diga_nchw_f32_to_nhwc_f32");
  return DIGA_OK;
}

extern "C" int diga_nhwc_f32_to_nchw_f32(const float* in, float* out, int B, int HW, int C, void* stream) {
  DIGA_CHECK_ARG(in && out && B > 0 && HW > 0 && C > 0, "diga_nhwc_f32_to_nchw_f32: bad argument");
  hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), B), dim3(256), 0, (hipStream_t)stream, in, out, HW, C);
  DIGA_CHECK_LAUNCH("diga_nhwc_f32_to_nchw_f32");
  return DIGA_OK;
}
