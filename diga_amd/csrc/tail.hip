// Full-resolution tail (DIGA.step ttda_module.py:1610 + SegmentationMetric.update ttda:158-173):
// bilinear upsample (align_corners=False) of the fused probabilities to the label resolution,
// arg-max over classes and confusion-matrix accumulation, fused so that the [B,C,H,W] fp32
// probabilities (40 MB per 1024x512 frame) never have to exist unless the caller asks for them.
#include "common.h"
#include <stdarg.h>

namespace {

thread_local char g_err[512] = "";

// PyTorch area_pixel_compute_source_index(scale, dst, align_corners=false, cubic=false)
__device__ __forceinline__ void src_index(int dst, int in, int out, int& i0, int& i1, float& l0, float& l1) {
  const float scale = (float)in / (float)out;
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}

// PLAIN: no interpolation at all (h == H, w == W): the metric called on full-resolution predictions, preds.argmax(1)
// (ttda:178-179).  The bilinear form with equal sizes would multiply the +1 neighbour by weight 0, and 0 * NaN = NaN would
// spread the 0/0 pixels of ttda:1733 to their left / upper neighbours -- which nn.Upsample does too, but argmax alone does not.
template <typename T, bool PLAIN = false>
__global__ __launch_bounds__(256) void upsample_argmax_hist_kernel(const float* __restrict__ probs, const T* __restrict__ target,
                                                                   uint8_t* __restrict__ pred, unsigned long long* __restrict__ hist,
                                                                   float* __restrict__ probs_up, int B, int C, int h, int w, int H,
                                                                   int W, int hist_scale) {
  __shared__ unsigned int sh[DIGA_NUM_CLASSES_MAX * DIGA_NUM_CLASSES_MAX];
  for (int e = threadIdx.x; e < C * C; e += 256) sh[e] = 0;
  __syncthreads();
  const int64_t total = (int64_t)B * H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % W), oy = (int)((i / W) % H), b = (int)(i / ((int64_t)W * H));
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    if constexpr (!PLAIN) {
      src_index(oy, h, H, y0, y1, ly0, ly1);
      src_index(ox, w, W, x0, x1, lx0, lx1);
    }
    const float* pb = probs + (size_t)b * C * h * w;
    float best = 0.f;
    int am = 0;
    for (int c = 0; c < C; ++c) {
      const float* pc = pb + (size_t)c * h * w;
      float v;
      if constexpr (PLAIN)
        v = pc[oy * w + ox];
      else
        v = ly0 * (lx0 * pc[y0 * w + x0] + lx1 * pc[y0 * w + x1]) + ly1 * (lx0 * pc[y1 * w + x0] + lx1 * pc[y1 * w + x1]);
      if (probs_up) probs_up[(((size_t)b * C + c) * H + oy) * W + ox] = v;
      const bool better = c == 0 || (v > best) || (v != v && best == best);
      if (better) {
        best = v;
        am = c;
      }
    }
    if (pred) pred[i] = (uint8_t)am;
    const long long t = (long long)target[i];
    if (t >= 0 && t < C) atomicAdd(&sh[(int)t * C + am], 1u);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < C * C; e += 256)
    if (sh[e]) atomicAdd(&hist[e], (unsigned long long)sh[e] * (unsigned long long)hist_scale);   // ttda:165: hist * batch_size
}

}  // namespace

void diga_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* diga_last_error_string(void) { return g_err; }
extern "C" int diga_version(void) { return 100; }

namespace {
int upsample_argmax_hist(const float* probs, const void* target, int target_is_i64, uint8_t* pred, int64_t* hist, float* probs_up, int B,
                         int C, int h, int w, int H, int W, int hist_scale, void* stream, bool plain = false) {
  DIGA_CHECK_ARG(probs && target && hist, "diga_upsample_argmax_hist: NULL pointer");
  DIGA_CHECK_ARG(B > 0 && C > 0 && C <= DIGA_NUM_CLASSES_MAX && h > 0 && w > 0 && H > 0 && W > 0, "diga_upsample_argmax_hist: bad sizes");
  const int64_t total = (int64_t)B * H * W;
  const int cap = 2048 * (B < 4 ? 1 : 4);
  const int grid = total / 256 < cap ? cdiv(total, 256) : cap;
  if (plain && target_is_i64)
    hipLaunchKernelGGL((upsample_argmax_hist_kernel<int64_t, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, probs,
                       (const int64_t*)target, pred, (unsigned long long*)hist, probs_up, B, C, h, w, H, W, hist_scale);
  else if (plain)
    hipLaunchKernelGGL((upsample_argmax_hist_kernel<int32_t, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, probs,
                       (const int32_t*)target, pred, (unsigned long long*)hist, probs_up, B, C, h, w, H, W, hist_scale);
  else if (target_is_i64)
    hipLaunchKernelGGL(upsample_argmax_hist_kernel<int64_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, probs,
                       (const int64_t*)target, pred, (unsigned long long*)hist, probs_up, B, C, h, w, H, W, hist_scale);
  else
    hipLaunchKernelGGL(upsample_argmax_hist_kernel<int32_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, probs,
                       (const int32_t*)target, pred, (unsigned long long*)hist, probs_up, B, C, h, w, H, W, hist_scale);
  DIGA_CHECK_LAUNCH("diga_upsample_argmax_hist");
  return DIGA_OK;
}
}  // namespace

extern "C" int diga_upsample_argmax_hist(const float* probs, const void* target, int target_is_i64, uint8_t* pred, int64_t* hist,
                                         float* probs_up, int B, int C, int h, int w, int H, int W, void* stream) {
  return upsample_argmax_hist(probs, target, target_is_i64, pred, hist, probs_up, B, C, h, w, H, W, B, stream);
}

extern "C" int diga_upsample_argmax_hist_frames(const float* probs, const void* target, int target_is_i64, uint8_t* pred, int64_t* hist,
                                                float* probs_up, int nframes, int C, int h, int w, int H, int W, void* stream) {
  return upsample_argmax_hist(probs, target, target_is_i64, pred, hist, probs_up, nframes, C, h, w, H, W, 1, stream);
}

extern "C" int diga_argmax_hist(const float* probs, const void* target, int target_is_i64, uint8_t* pred, int64_t* hist, int B, int C,
                                int H, int W, void* stream) {
  return upsample_argmax_hist(probs, target, target_is_i64, pred, hist, nullptr, B, C, H, W, H, W, B, stream, true);
}
