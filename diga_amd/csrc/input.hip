// Input pipeline contract: what the reference's datasets do between the decoded image and the
// network input, on the GPU.
//   BaseDataset.preprocess (base_dataset.py:45-52):  image[:, :, ::-1] (RGB -> BGR), image -= mean, HWC -> CHW.
//     `image` is float32, `mean` float64: numpy evaluates the in-place subtraction in float64 and casts back, so the
//     kernel does exactly that: out = (float)((double)pixel - mean[c]).
//   BDD100K / Mapillary (bdd100k.py:290, mapillary.py): np.array(img) - mean WITHOUT the channel swap (the BGR mean is
//     subtracted from RGB data), float64, then .float(): the same arithmetic with swap_rb = 0.
//   CityscapesDataSet.map_labels (cityscapes.py:50-51):  map_vector[label.astype(int64)] with numpy's negative
//     indexing (id -1 -> last entry).
// Both kernels are pure streaming (1 read, 1 write): HBM-bound.
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void preprocess_kernel(const T* __restrict__ img, float* __restrict__ out, int64_t HW, double m0,
                                                         double m1, double m2, int swap_rb) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const int64_t b = blockIdx.y;
  const T* px = img + (b * HW + p) * 3;
  const double v0 = (double)px[0], v1 = (double)px[1], v2 = (double)px[2];
  float* o = out + b * 3 * HW + p;
  o[0] = (float)((swap_rb ? v2 : v0) - m0);
  o[HW] = (float)(v1 - m1);
  o[2 * HW] = (float)((swap_rb ? v0 : v2) - m2);
}

template <typename T, typename O>
__global__ __launch_bounds__(256) void map_labels_kernel(const T* __restrict__ ids, const int32_t* __restrict__ lut, int nlut,
                                                         O* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  long long id = (long long)ids[i];                 // float -> int64 truncation, as ndarray.astype(np.int64)
  if (id < 0) id += nlut;                           // numpy negative indexing
  out[i] = (id >= 0 && id < nlut) ? (O)lut[id] : (O)255;   // numpy would raise IndexError: mapped to "ignore"
}

}  // namespace

extern "C" int diga_preprocess_image(const void* img, int img_is_u8, float* out, int B, int H, int W, double mean0, double mean1,
                                     double mean2, int swap_rb, void* stream) {
  DIGA_CHECK_ARG(img && out && B > 0 && H > 0 && W > 0, "diga_preprocess_image: bad argument");
  const int64_t HW = (int64_t)H * W;
  const dim3 grid(cdiv(HW, 256), B), block(256);
  if (img_is_u8)
    hipLaunchKernelGGL(preprocess_kernel<uint8_t>, grid, block, 0, (hipStream_t)stream, (const uint8_t*)img, out, HW, mean0, mean1,
                       mean2, swap_rb);
  else
    hipLaunchKernelGGL(preprocess_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)img, out, HW, mean0, mean1, mean2,
                       swap_rb);
  DIGA_CHECK_LAUNCH("diga_preprocess_image");
  return DIGA_OK;
}

extern "C" int diga_map_labels(const void* ids, int ids_kind, const int32_t* lut, int nlut, void* out, int out_is_i64, int64_t n,
                               void* stream) {
  DIGA_CHECK_ARG(ids && lut && out && n > 0 && nlut > 0 && ids_kind >= 0 && ids_kind <= 2, "diga_map_labels: bad argument");
  const dim3 grid(cdiv(n, 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define DIGA_ML(T, O) hipLaunchKernelGGL((map_labels_kernel<T, O>), grid, block, 0, st, (const T*)ids, lut, nlut, (O*)out, n)
  if (out_is_i64) {
    if (ids_kind == 0) DIGA_ML(uint8_t, int64_t);
    else if (ids_kind == 1) DIGA_ML(int32_t, int64_t);
    else DIGA_ML(float, int64_t);
  } else {
    if (ids_kind == 0) DIGA_ML(uint8_t, int32_t);
    else if (ids_kind == 1) DIGA_ML(int32_t, int32_t);
    else DIGA_ML(float, int32_t);
  }
#undef DIGA_ML
  DIGA_CHECK_LAUNCH("diga_map_labels");
  return DIGA_OK;
}
