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

// The same arithmetic from an LDS copy of the source patch (up-sampling by a factor: the 8 x 128 output pixels of a tile read
// a few source rows and columns 4 x C times each).  A block walks tiles; the patch of a tile is rows y0(first row) ... y1(last
// row) x columns x0(first column) ... x1(last column) of every class plane.  The launch code uses this form when the patch of
// every tile fits TL_PH x TL_PW cells (src_index is monotone, so the corner pixels bound it).
constexpr int TL_W = 128, TL_H = 8, TL_PW = 24, TL_PH = 6;

template <typename T>
__global__ __launch_bounds__(256) void upsample_argmax_hist_tiled_kernel(const float* __restrict__ probs, const T* __restrict__ target,
                                                                         uint8_t* __restrict__ pred, unsigned long long* __restrict__ hist,
                                                                         float* __restrict__ probs_up, int B, int C, int h, int w, int H,
                                                                         int W, int hist_scale, int tiles_x, int tiles_y) {
  __shared__ unsigned int sh[DIGA_NUM_CLASSES_MAX * DIGA_NUM_CLASSES_MAX];
  __shared__ float sp[DIGA_NUM_CLASSES_MAX * TL_PH * TL_PW];
  for (int e = threadIdx.x; e < C * C; e += 256) sh[e] = 0;
  const int ntiles = B * tiles_y * tiles_x;
  const int lx = threadIdx.x & (TL_W - 1), ly = threadIdx.x >> 7;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t = tile;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int ox0 = tx * TL_W, oy0 = ty * TL_H;
    int ylo, yhi, xlo, xhi, tmp;
    float f0, f1;
    src_index(oy0, h, H, ylo, tmp, f0, f1);
    src_index(min(oy0 + TL_H, H) - 1, h, H, tmp, yhi, f0, f1);
    src_index(ox0, w, W, xlo, tmp, f0, f1);
    src_index(min(ox0 + TL_W, W) - 1, w, W, tmp, xhi, f0, f1);
    const int ph = yhi - ylo + 1, pw = xhi - xlo + 1, cells = ph * pw;
    const float* pb = probs + (size_t)b * C * h * w;
    __syncthreads();             // the previous tile's patch has been read (first pass: the histogram is zero)
    for (int e = threadIdx.x; e < C * cells; e += 256) {
      const int c = e / cells, r = e - c * cells, py = r / pw, px = r - py * pw;
      sp[e] = pb[((size_t)c * h + ylo + py) * w + xlo + px];
    }
    __syncthreads();
    const int ox = ox0 + lx;
#pragma unroll
    for (int k = 0; k < TL_H / 2; ++k) {
      const int oy = oy0 + 2 * k + ly;
      if (ox >= W || oy >= H) continue;
      int y0, y1, x0, x1;
      float ly0, ly1, lx0, lx1;
      src_index(oy, h, H, y0, y1, ly0, ly1);
      src_index(ox, w, W, x0, x1, lx0, lx1);
      const int o00 = (y0 - ylo) * pw + (x0 - xlo), o01 = (y0 - ylo) * pw + (x1 - xlo);
      const int o10 = (y1 - ylo) * pw + (x0 - xlo), o11 = (y1 - ylo) * pw + (x1 - xlo);
      float best = 0.f;
      int am = 0;
      for (int c = 0; c < C; ++c) {
        const float* pc = sp + c * cells;
        const float v = ly0 * (lx0 * pc[o00] + lx1 * pc[o01]) + ly1 * (lx0 * pc[o10] + lx1 * pc[o11]);
        if (probs_up) probs_up[(((size_t)b * C + c) * H + oy) * W + ox] = v;
        const bool better = c == 0 || (v > best) || (v != v && best == best);
        if (better) {
          best = v;
          am = c;
        }
      }
      const int64_t i = ((int64_t)b * H + oy) * W + ox;
      if (pred) pred[i] = (uint8_t)am;
      const long long tg = (long long)target[i];
      if (tg >= 0 && tg < C) atomicAdd(&sh[(int)tg * C + am], 1u);
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < C * C; e += 256)
    if (sh[e]) atomicAdd(&hist[e], (unsigned long long)sh[e] * (unsigned long long)hist_scale);
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
  // up-sampling by a factor: the tiled form, if the source patch of an 8 x 128 output tile is sure to fit its LDS image
  if (!plain && (int64_t)TL_H * h / H + 3 <= TL_PH && (int64_t)TL_W * w / W + 3 <= TL_PW) {
    const int tiles_x = cdiv(W, TL_W), tiles_y = cdiv(H, TL_H);
    const int64_t ntiles = (int64_t)B * tiles_x * tiles_y;
    const int tgrid = ntiles < cap ? (int)ntiles : cap;
    if (target_is_i64)
      hipLaunchKernelGGL(upsample_argmax_hist_tiled_kernel<int64_t>, dim3(tgrid), dim3(256), 0, (hipStream_t)stream, probs,
                         (const int64_t*)target, pred, (unsigned long long*)hist, probs_up, B, C, h, w, H, W, hist_scale, tiles_x, tiles_y);
    else
      hipLaunchKernelGGL(upsample_argmax_hist_tiled_kernel<int32_t>, dim3(tgrid), dim3(256), 0, (hipStream_t)stream, probs,
                         (const int32_t*)target, pred, (unsigned long long*)hist, probs_up, B, C, h, w, H, W, hist_scale, tiles_x, tiles_y);
    DIGA_CHECK_LAUNCH("diga_upsample_argmax_hist");
    return DIGA_OK;
  }
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
