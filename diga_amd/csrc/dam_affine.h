// The arithmetic that turns a BN layer's group sums into its DAM affine (ttda_module.py:313-343), shared by
// diga_bn_finalize / diga_bn_finalize_sums (bn.hip) and the convolution kernel's fused finalize (conv_igemm.hip) so that
// every path produces bit-identical scale / shift.
#pragma once
#include "common.h"

// Slab = [unit][2 (sum, sumsq)][C]; a unit range is summed in FIN_PH interleaved phases (phase w takes units u0+w,
// u0+w+FIN_PH, ...) in fp64, the phases are then added in phase order: a fixed order, independent of who computes it.
constexpr int FIN_PH = 16;

// COHERENT: the slab was written by other blocks of the SAME launch, possibly through another XCD's L2: read it with
// agent-scope loads (they bypass this XCD's possibly stale L2 lines) instead of invalidating the whole L2.
template <bool COHERENT = false>
__device__ __forceinline__ void slab_range_sums(const float* __restrict__ slab, int u0, int u1, int C, int c, int ph, double* s) {
  s[0] = s[1] = 0;
#pragma unroll 4
  for (int u = u0 + ph; u < u1; u += FIN_PH) {
    const float* q = slab + (size_t)u * 2 * C + c;
    if constexpr (COHERENT) {
      s[0] += (double)__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s[1] += (double)__hip_atomic_load(q + C, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      s[0] += (double)q[0];
      s[1] += (double)q[C];
    }
  }
}

// sums of group A (units [a0,a1)) and group B (units [a1,b1)) -> s[4] valid in phase 0 of the block
__device__ __forceinline__ void slab_group_sums(const float* __restrict__ slab, int a0, int a1, int b1, int C, int c, bool live,
                                                double (*red)[4][64], double* s) {
  const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
  s[0] = s[1] = s[2] = s[3] = 0;
  if (live) {
    slab_range_sums(slab, a0, a1, C, c, ph, s);
    slab_range_sums(slab, a1, b1, C, c, ph, s + 2);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) red[ph][k][lane] = s[k];
  __syncthreads();
  if (ph != 0) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double v = 0;
    for (int w = 0; w < FIN_PH; ++w) v += red[w][k][lane];
    s[k] = v;
  }
}

// group sums (fp64) -> lambda-mixed scale/shift of channel c  (ttda:313-343)
__device__ __forceinline__ void dam_affine_from_sums(const double* s, int c, double n_a, double n_b, const float* __restrict__ rmean,
                                                     const float* __restrict__ rvar, const float* __restrict__ weight,
                                                     const float* __restrict__ bias, float lam, float eps,
                                                     float* __restrict__ scale, float* __restrict__ shift, int out_index = -1) {
  // out_index >= 0: the affine goes to scale / shift [out_index] instead of [c] (a block-local array)
  const int oc = out_index >= 0 ? out_index : c;
  const double K = (double)rmean[c];
  const double ma = s[0] / n_a;
  float mean_cur = (float)(K + ma);
  float var_cur = (float)(s[1] / n_a - ma * ma);
  if (n_b > 0) {
    const double mb = s[2] / n_b;
    const float mean_b = (float)(K + mb);
    const float var_b = (float)(s[3] / n_b - mb * mb);
    mean_cur = (mean_cur + mean_b) / 2.f;   // ttda:316-317 (fp32)
    var_cur = (var_cur + var_b) / 2.f;
  }
  const float oml = 1.f - lam;              // fp32 subtraction, as the reference's tensor op (ttda:335)
  const float mean = lam * rmean[c] + oml * mean_cur;
  const float var = lam * rvar[c] + oml * var_cur;
  const float sc = weight[c] / sqrtf(var + eps);
  scale[oc] = sc;
  shift[oc] = bias[c] - mean * sc;
}
