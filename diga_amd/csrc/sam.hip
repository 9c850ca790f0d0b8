// Semantic Adaptation Module kernels (ttda_module.py:1662-1805) and the fusion line (ttda:1644).
#include "common.h"
#include <mutex>

namespace {

// ---------------------------------------------------------------------------------------------
// head finish + pseudo labels  (ttda:1662-1676)
// One thread per pixel: logits = sum of split-K partials + (bias_d6 + bias_d12); softmax over the
// C classes in fixed class order; label = argmax (lowest index on ties, NaN counts as maximal, as
// torch.argmax does) and 255 where max prob < threshold (strict).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_finish_pl_kernel(const float* __restrict__ partials, int splitk, int cpad,
                                                             const float* __restrict__ bias, const float* __restrict__ logits_in,
                                                             float* __restrict__ logits_out, int32_t* __restrict__ labels,
                                                             int64_t P, int C, int hw, float thr, int64_t lstride) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int64_t b = p / hw, pix = p - b * hw;
  float l[DIGA_NUM_CLASSES_MAX];
  if (partials) {
    for (int c = 0; c < C; ++c) {
      float v = 0.f;
      for (int s = 0; s < splitk; ++s) v += partials[((size_t)s * P + p) * cpad + c];
      l[c] = v + bias[c];
    }
  } else {
    for (int c = 0; c < C; ++c) l[c] = logits_in[((size_t)b * C + c) * hw + pix];
  }
  if (logits_out)
    for (int c = 0; c < C; ++c) logits_out[((size_t)b * C + c) * hw + pix] = l[c];
  float mx = l[0];
  int am = 0;
  for (int c = 1; c < C; ++c) {
    const bool better = (l[c] > mx) || (l[c] != l[c] && mx == mx);
    if (better) {
      mx = l[c];
      am = c;
    }
  }
  float sum = 0.f;
  for (int c = 0; c < C; ++c) sum += expf(l[c] - mx);
  const float pmax = 1.0f / sum;        // exp(0)/sum
  labels[b * lstride + pix] = (pmax < thr) ? 255 : am;  // NaN < thr is false -> keeps argmax, as the reference
}

// ---------------------------------------------------------------------------------------------
// head as ONE dense GEMM + tap gather (adaptseg:119-121): z[m][t*C + c] = feat[m] . W_t[c] for the 18 taps of the two
// executed branches is a plain 1x1 igemm (the feature map is read once instead of 18 times); the 3x3 dilated
// convolutions are then logits[m][c] = bias[c] + sum_t z[m + shift_t][t*C + c] over the taps that land inside the
// image (zero padding).  32 lanes per pixel (lane = class): every tap is one contiguous 76-byte read per pixel.
// The label rule is the one of head_finish_pl_kernel (same arithmetic order).
// ---------------------------------------------------------------------------------------------
struct TapList {
  int n;
  short dy[DIGA_MAX_TAPS], dx[DIGA_MAX_TAPS];
};

constexpr int HTS_PX = 64;      // pixels per block

// Block = 64 pixels.  Phase 1: 8 groups of 32 lanes (lane = class) take 8 pixels each; the tap reads of a pixel are issued in
// batches of 6 before they are summed (in tap order: the arithmetic of the one-pixel-per-group form), sums -> LDS.  Phase 2: the
// logits are written with the PIXEL as the fast index (NCHW planes: 256-byte runs instead of 4-byte scatters).  Phase 3: one thread
// per pixel applies the label rule.
__global__ __launch_bounds__(256) void head_tapsum_pl_kernel(const float* __restrict__ z, int zstride, const TapList taps,
                                                             const float* __restrict__ bias, float* __restrict__ logits_out,
                                                             int32_t* __restrict__ labels, int64_t P, int C, int h, int w, float thr,
                                                             int64_t lstride) {
  __shared__ float sl[HTS_PX][33];
  const int lane = threadIdx.x & 31, pp = threadIdx.x >> 5;
  const int64_t p0 = (int64_t)blockIdx.x * HTS_PX;
  const int hw = h * w;
  const float bv = lane < C ? bias[lane] : 0.f;
  for (int g = 0; g < HTS_PX / 8; ++g) {
    const int lp = g * 8 + pp;
    const int64_t p = p0 + lp;
    float acc = bv;
    if (p < P && lane < C) {
      const int64_t b = p / hw;
      const int pix = (int)(p - b * hw);
      const int oh = pix / w, ow = pix - oh * w;
      const float* zb = z + (size_t)(b * hw) * zstride + lane;
#pragma unroll
      for (int t0 = 0; t0 < DIGA_MAX_TAPS; t0 += 6) {
        float v[6];
        bool ok[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int t = t0 + j;
          const int ih = oh + taps.dy[t], iw = ow + taps.dx[t];
          ok[j] = t < taps.n && (unsigned)ih < (unsigned)h && (unsigned)iw < (unsigned)w;
          v[j] = ok[j] ? zb[(size_t)(ih * w + iw) * zstride + t * C] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j)
          if (ok[j]) acc += v[j];
      }
    }
    sl[lp][lane] = acc;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < HTS_PX * C; e += 256) {
    const int c = e / HTS_PX, lp = e - c * HTS_PX;
    const int64_t p = p0 + lp;
    if (p < P) {
      const int64_t b = p / hw;
      logits_out[((size_t)b * C + c) * hw + (p - b * hw)] = sl[lp][c];
    }
  }
  const int64_t p = p0 + threadIdx.x;
  if (threadIdx.x >= HTS_PX || p >= P) return;
  const float* l = sl[threadIdx.x];
  float mx = l[0];
  int am = 0;
  for (int c = 1; c < C; ++c) {
    const float v = l[c];
    const bool better = (v > mx) || (v != v && mx == mx);
    if (better) {
      mx = v;
      am = c;
    }
  }
  float sum = 0.f;
  for (int c = 0; c < C; ++c) sum += expf(l[c] - mx);
  const float pmax = 1.0f / sum;
  const int64_t b = p / hw;
  labels[b * lstride + (p - b * hw)] = (pmax < thr) ? 255 : am;
}

// ---------------------------------------------------------------------------------------------
// nearest down-sampling of the label map (ttda:1695)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
  if (in == out) return dst;
  if (in * 2 == out) return dst >> 1;
  const float scale = (float)in / (float)out;
  const int s = (int)floorf((float)dst * scale);
  return s < in - 1 ? s : in - 1;
}

template <typename T>
__global__ void nearest_down_kernel(const T* __restrict__ y, int32_t* __restrict__ out, int B, int H, int W, int h, int w,
                                    int64_t ostride) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * h * w) return;
  const int ox = (int)(i % w), oy = (int)((i / w) % h), b = (int)(i / ((int64_t)w * h));
  out[b * ostride + (int64_t)oy * w + ox] = (int32_t)y[((size_t)b * H + nearest_src(oy, H, h)) * W + nearest_src(ox, W, w)];
}

// ---------------------------------------------------------------------------------------------
// prototype pooling (cal_proto, ttda:1788-1805)
//   1. index: one wave per (set, class) walks the label array in pixel order and writes the stable list
//      of pixel ids of that class (ballot + popcount compaction)                       -> idx, counts
//   2. pool : block (chunk j, class c, set s): each wave owns a 512-channel slice (8 per lane) and sums the
//      feature rows of list positions [j*CS, (j+1)*CS) in registers                   -> partial[s][c][j][CH]
//   3. reduce: sums[s][c][ch] = sum_j partial (fixed order)                            -> deterministic
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void proto_index_kernel(const int32_t* __restrict__ labels, int64_t P, int C,
                                                         int32_t* __restrict__ idx, int32_t* __restrict__ counts) {
  const int c = blockIdx.x, s = blockIdx.y, lane = threadIdx.x;
  const int32_t* lab = labels + (size_t)s * P;
  int32_t* list = idx + ((size_t)s * C + c) * P;
  int count = 0;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int64_t base = 0; base < P; base += 256) {
    int v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t p = base + lane * 4 + j;
      v[j] = p < P ? lab[p] : -1;
    }
    unsigned long long m[4];
    int before = 0, total = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      m[j] = __ballot(v[j] == c);
      before += __popcll(m[j] & below);
      total += __popcll(m[j]);
    }
    int pos = count + before;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (v[j] == c) list[pos++] = (int32_t)(base + lane * 4 + j);
    }
    count += total;
  }
  if (lane == 0) counts[s * C + c] = count;
}

// 8 consecutive channels of one feature row as fp32: bf16 features (the product path) or fp32 features (the verification
// mode and the helper API called with arbitrary fp32 tensors, as the reference computes, ttda:1788-1805)
__device__ __forceinline__ void load_row8(const unsigned short* p, float* f) { unpack_bf16x8(*reinterpret_cast<const u32x4*>(p), f); }
__device__ __forceinline__ void load_row8(const float* p, float* f) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  f[0] = a[0], f[1] = a[1], f[2] = a[2], f[3] = a[3], f[4] = b[0], f[5] = b[1], f[6] = b[2], f[7] = b[3];
}

template <typename FT>
__global__ __launch_bounds__(256) void proto_pool_kernel(const FT* __restrict__ feat, const int32_t* __restrict__ idx,
                                                         const int32_t* __restrict__ counts, int64_t P, int C, int CH, int CS,
                                                         int JC, float* __restrict__ partial, int nsets) {
  const int j = blockIdx.x, c = blockIdx.y, s = blockIdx.z;      // s = frame * nsets + set: every frame has its own features
  feat += (size_t)(s / nsets) * P * CH;
  const int n = counts[s * C + c];
  const int beg = j * CS;
  if (beg >= n) return;
  const int end = min(beg + CS, n);
  const int32_t* list = idx + ((size_t)s * C + c) * P;
  // chunks are laid out compactly per set: class c's chunks follow those of the classes before it (at most JC + C per set)
  int off = 0;
  for (int k = 0; k < c; ++k) off += (counts[s * C + k] + CS - 1) / CS;
  float* dst = partial + ((size_t)s * (JC + C) + off + j) * CH;
  for (int ch0 = threadIdx.x * 8; ch0 < CH; ch0 += 2048) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    int q = beg;
    for (; q + 8 <= end; q += 8) {        // 8 independent row reads in flight per lane
      float f[8][8];
#pragma unroll
      for (int u = 0; u < 8; ++u) load_row8(feat + (size_t)list[q + u] * CH + ch0, f[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += f[u][i];
      }
    }
    for (; q < end; ++q) {
      float f[8];
      load_row8(feat + (size_t)list[q] * CH + ch0, f);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += f[i];
    }
    *reinterpret_cast<f32x4*>(dst + ch0) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *reinterpret_cast<f32x4*>(dst + ch0 + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
  }
}

__global__ void proto_reduce_kernel(const float* __restrict__ partial, const int32_t* __restrict__ counts, int C, int CH, int CS,
                                    int JC, float* __restrict__ sums) {
  const int sc = blockIdx.y;                                   // s*C + c
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= CH) return;
  const int n = counts[sc];
  const int nj = (n + CS - 1) / CS;
  const int s = sc / C, c = sc - s * C;
  int off = 0;
  for (int k = 0; k < c; ++k) off += (counts[s * C + k] + CS - 1) / CS;
  float v = 0.f;
  for (int j = 0; j < nj; ++j) v += partial[((size_t)s * (JC + C) + off + j) * CH + ch];
  sums[(size_t)sc * CH + ch] = v;
}

// ---------------------------------------------------------------------------------------------
// instance prototypes + running-prototype seed / EMA (ttda:1700-1705, 1736-1765)
// ---------------------------------------------------------------------------------------------
__global__ void proto_update_kernel(const float* __restrict__ sums, const int32_t* __restrict__ counts, int C, int CH,
                                    float* __restrict__ inst_proto, float* __restrict__ inst_flag, float* __restrict__ run_proto,
                                    float* __restrict__ run_flag, const int32_t* __restrict__ state_flag, float omr, float rho) {
  const int c = blockIdx.y;
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= CH) return;
  const int n_pl = counts[c], n_gt = counts[C + c];
  const float inst = n_pl > 0 ? sums[(size_t)c * CH + ch] / (float)n_pl : 0.f;   // .mean(0): sum / n
  inst_proto[(size_t)c * CH + ch] = inst;
  const bool seeded = *state_flag != 0;
  float run = run_proto[(size_t)c * CH + ch];
  const float rf = run_flag[c];
  float new_flag = rf;
  if (!seeded) {                      // first frame: running := instance (ttda:1702-1703)
    run = inst;
    new_flag = n_pl > 0 ? 1.f : 0.f;
  } else if (n_gt > 0 && rho != 0.f) {
    const float gt = sums[((size_t)C + c) * CH + ch] / (float)n_gt;
    if (rf == 0.f) {
      run = gt;                       // first time the class is seen (ttda:1760-1762)
      new_flag = 1.f;
    } else {
      run = omr * run + rho * gt;     // ttda:1764
    }
  }
  run_proto[(size_t)c * CH + ch] = run;
  // flags are written by a second tiny kernel (below) so that no block reads a flag another block already updated
}

__global__ void proto_flags_kernel(const int32_t* __restrict__ counts, int C, float* __restrict__ inst_flag,
                                   float* __restrict__ run_flag, int32_t* __restrict__ state_flag, float rho) {
  const int c = threadIdx.x;
  const bool seeded = *state_flag != 0;
  __syncthreads();
  if (c < C) {
    const int n_pl = counts[c], n_gt = counts[C + c];
    inst_flag[c] = n_pl > 0 ? 1.f : 0.f;
    if (!seeded)
      run_flag[c] = n_pl > 0 ? 1.f : 0.f;
    else if (n_gt > 0 && rho != 0.f)
      run_flag[c] = 1.f;
  }
  if (c == 0) *state_flag = 1;
}

// The same chain for F consecutive batch-1 frames of a stream in ONE launch pair: thread (c, ch) walks the frames in
// order with the running value in a register (flags and the seeded state evolve identically in every thread: they
// depend on the counts only) and stores, per frame, the instance prototype and the running prototype AFTER that
// frame's update -- what frame f's classifier reads.  Arithmetic per frame = proto_update_kernel's.
__global__ void proto_update_frames_kernel(const float* __restrict__ sums, const int32_t* __restrict__ counts, int F, int C, int CH,
                                           float* __restrict__ inst_protos, float* __restrict__ run_protos,
                                           float* __restrict__ run_proto, const float* __restrict__ run_flag,
                                           const int32_t* __restrict__ state_flag, float omr, float rho) {
  const int c = blockIdx.y;
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= CH) return;
  bool seeded = *state_flag != 0;
  float run = run_proto[(size_t)c * CH + ch];
  float rf = run_flag[c];
  for (int f = 0; f < F; ++f) {
    const int32_t* cnt = counts + (size_t)f * 2 * C;
    const float* sm = sums + (size_t)f * 2 * C * CH;
    const int n_pl = cnt[c], n_gt = cnt[C + c];
    const float inst = n_pl > 0 ? sm[(size_t)c * CH + ch] / (float)n_pl : 0.f;
    inst_protos[((size_t)f * C + c) * CH + ch] = inst;
    if (!seeded) {
      run = inst;
      rf = n_pl > 0 ? 1.f : 0.f;
    } else if (n_gt > 0 && rho != 0.f) {
      const float gt = sm[((size_t)C + c) * CH + ch] / (float)n_gt;
      if (rf == 0.f) {
        run = gt;
        rf = 1.f;
      } else {
        run = omr * run + rho * gt;
      }
    }
    seeded = true;
    run_protos[((size_t)f * C + c) * CH + ch] = run;
  }
  run_proto[(size_t)c * CH + ch] = run;
}

__global__ void proto_flags_frames_kernel(const int32_t* __restrict__ counts, int F, int C, float* __restrict__ inst_flags,
                                          float* __restrict__ run_flags, float* __restrict__ run_flag,
                                          int32_t* __restrict__ state_flag, float rho) {
  const int c = threadIdx.x;
  bool seeded = *state_flag != 0;
  __syncthreads();
  if (c < C) {
    float rf = run_flag[c];
    for (int f = 0; f < F; ++f) {
      const int n_pl = counts[(size_t)f * 2 * C + c], n_gt = counts[(size_t)f * 2 * C + C + c];
      inst_flags[f * C + c] = n_pl > 0 ? 1.f : 0.f;
      if (!seeded)
        rf = n_pl > 0 ? 1.f : 0.f;
      else if (n_gt > 0 && rho != 0.f)
        rf = 1.f;
      seeded = true;
      run_flags[f * C + c] = rf;
    }
    run_flag[c] = rf;
  }
  if (c == 0) *state_flag = 1;
}

// ---------------------------------------------------------------------------------------------
// prototype-distance classifier x2 + blend + softmax fusion (ttda:1707-1709, 1712-1734, 1644)
// Squared distances are accumulated directly, sum (p - f)^2 in fp32 -- no ||f||^2 - 2 f.p + ||p||^2 expansion
// (an absolute error in d is a relative error in exp(-d)) -- so this is fp32 VALU work: P x 2C x CH x 2 ops.
//   1. proto_dist_kernel: block = 128 pixels x 40 prototype slots (slot k < C: instance class k, C <= k < 2C:
//      running class k-C) x one slice of CH/PCF_SPLIT channels, 32 channels at a time through LDS; 128 threads,
//      thread = 8 pixels x 5 slots in registers: per channel two b128 reads of 8 pixel values ([ch][px] image) and one
//      b128 + b32 broadcast read of 5 prototype values feed 40 packed VALU ops.  Partial d^2 -> work[split][P][40].
//   2. proto_fuse_kernel: thread = pixel: d^2 = sum of the slices in fixed order, exp(-d), flags, normalise x2,
//      blends, softmax fusion; 0/0 -> NaN kept.
// ---------------------------------------------------------------------------------------------
constexpr int PCF_PIX = 128;
constexpr int PCF_CHUNK = 32;     // channels per LDS image
constexpr int PCF_SLOTS = 40;
constexpr int PCF_SPLIT = 8;
constexpr int PCF_PITCH = 132;    // floats per channel row of the pixel image (128 + 4: keeps b128 alignment)

// prototypes -> protoT[ch][8 slot groups][8] fp32 (5 used per group): the image proto_dist_kernel copies into LDS
__global__ void proto_transpose_kernel(const float* __restrict__ inst_proto, const float* __restrict__ run_proto,
                                       float* __restrict__ protoT, int C, int CH) {
  const int e = blockIdx.x * 256 + threadIdx.x;          // one thread per (ch, padded slot)
  if (e >= CH * 64) return;
  inst_proto += (size_t)blockIdx.y * C * CH;             // blockIdx.y = prototype set (one per frame of a stream in flight)
  run_proto += (size_t)blockIdx.y * C * CH;
  protoT += (size_t)blockIdx.y * CH * 64;
  const int ch = e >> 6, g = (e >> 3) & 7, j = e & 7, k = g * 5 + j;
  float v = 0.f;
  if (j < 5) {
    if (k < C)
      v = inst_proto[(size_t)k * CH + ch];
    else if (k < 2 * C)
      v = run_proto[(size_t)(k - C) * CH + ch];
  }
  protoT[e] = v;
}

template <typename FT>
__global__ __launch_bounds__(128) void proto_dist_kernel(const FT* __restrict__ feat, const float* __restrict__ protoT,
                                                         float* __restrict__ work, int64_t P, int CH, int ch_per_split, int hw,
                                                         int bps, int64_t proto_set_stride) {
  __shared__ __attribute__((aligned(16))) float sf[PCF_CHUNK][PCF_PITCH];   // [ch][px]
  __shared__ __attribute__((aligned(16))) float spr[PCF_CHUNK][8][8];       // [ch][slot group][5 used of 8]
  const int tid = threadIdx.x, pg = tid & 15, sg = tid >> 4;
  // blocks never straddle samples (bps blocks per sample), so that every sample can have its own prototypes
  const int smp = blockIdx.x / bps, pix0 = (blockIdx.x - smp * bps) * PCF_PIX;
  const int64_t p0 = (int64_t)smp * hw + pix0;
  protoT += (size_t)smp * proto_set_stride;
  const int chb = blockIdx.y * ch_per_split;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 acc[4][5];                       // [pixel pair][prototype]: packed fp32 (v_pk_add_f32 / v_pk_fma_f32)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 5; ++k) acc[i][k] = f32x2{0.f, 0.f};
  // the next chunk's global data (4 feature octets + 8 prototype float4 per thread) is fetched into registers while
  // the current chunk is being reduced, so its latency is off the critical path
  float fr[4][8];
  f32x4 pr[4];
  auto fetch = [&](int ch0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {            // feature tile: 128 px x 4 octets, 4 per thread; 4 lanes read one row's 32 channels
      const int id = tid + 128 * j, r = id >> 2, o = id & 3;
      if (pix0 + r < hw) {
        load_row8(feat + (size_t)(p0 + r) * CH + ch0 + o * 8, fr[j]);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) fr[j][i] = 0.f;
      }
    }
    const f32x4* src = reinterpret_cast<const f32x4*>(protoT + (size_t)ch0 * 64);   // prototype image: straight 8 KB copy
#pragma unroll
    for (int j = 0; j < 4; ++j) pr[j] = src[tid + 128 * j];
  };
  fetch(chb);
  for (int ch0 = chb; ch0 < chb + ch_per_split; ch0 += PCF_CHUNK) {
    __syncthreads();                           // everyone is done reading the previous chunk's LDS image
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int id = tid + 128 * j, r = id >> 2, o = id & 3;
#pragma unroll
      for (int i = 0; i < 8; ++i) sf[o * 8 + i][r] = fr[j][i];
    }
    {
      f32x4* dst = reinterpret_cast<f32x4*>(&spr[0][0][0]);
#pragma unroll
      for (int j = 0; j < 4; ++j) dst[tid + 128 * j] = pr[j];
    }
    __syncthreads();
    if (ch0 + PCF_CHUNK < chb + ch_per_split) fetch(ch0 + PCF_CHUNK);
#pragma unroll 4
    for (int ch = 0; ch < PCF_CHUNK; ++ch) {
      const f32x4 fa = *reinterpret_cast<const f32x4*>(&sf[ch][pg * 8]);
      const f32x4 fb = *reinterpret_cast<const f32x4*>(&sf[ch][pg * 8 + 4]);
      const f32x4 q = *reinterpret_cast<const f32x4*>(&spr[ch][sg][0]);
      const float q4 = spr[ch][sg][4];
      const f32x2 fp[4] = {f32x2{fa[0], fa[1]}, f32x2{fa[2], fa[3]}, f32x2{fb[0], fb[1]}, f32x2{fb[2], fb[3]}};
      const float qq[5] = {q[0], q[1], q[2], q[3], q4};
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const f32x2 qk = f32x2{qq[k], qq[k]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x2 d = qk - fp[i];
          acc[i][k] = __builtin_elementwise_fma(d, d, acc[i][k]);
        }
      }
    }
  }
  float* w = work + (size_t)blockIdx.y * P * PCF_SLOTS;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t p = p0 + pg * 8 + i;
    if (pix0 + pg * 8 + i < hw) {
#pragma unroll
      for (int k = 0; k < 5; ++k) w[p * PCF_SLOTS + sg * 5 + k] = acc[i >> 1][k][i & 1];
    }
  }
}

// Block = PFU_PX pixels.  Phase 1: items (pixel, slot), contiguous in the partial-distance array, sum the channel slices of a
// squared distance in slice order and form exp(-d/tau)*flag; items (class, pixel) fetch the logits with the PIXEL as the fast
// index (128-byte runs of the NCHW planes).  Phase 2: one thread per pixel forms the three normalisers over the classes in
// fixed order.  Phase 3: items (class, pixel) write the fused probability, again in 128-byte runs.  (With 6 pixels per block
// and a thread per (pixel, class) the plane accesses were 24-byte pieces: 87 us per 15 frames against 62 here.)
constexpr int PFU_PX = 32;

__global__ __launch_bounds__(256) void proto_fuse_kernel(const float* __restrict__ work, const float* __restrict__ logits,
                                                         const float* __restrict__ inst_flag, const float* __restrict__ run_flag,
                                                         float* __restrict__ out, int64_t P, int C, int hw, int nsplit, float tau,
                                                         float pl, float pl1, float fl, float fl1, int flag_set_stride) {
  __shared__ float sw[PFU_PX][PCF_SLOTS + 1];
  __shared__ float sl[PFU_PX][DIGA_NUM_CLASSES_MAX + 1];
  __shared__ float sn[PFU_PX][4];                          // si, sr, mx, se of a pixel
  const int tid = threadIdx.x;
  const int64_t p0 = (int64_t)blockIdx.x * PFU_PX;
  for (int e = tid; e < PFU_PX * PCF_SLOTS; e += 256) {
    const int px = e / PCF_SLOTS, k = e - px * PCF_SLOTS;
    const int64_t p = p0 + px;
    float w = 0.f;
    if (p < P && k < 2 * C) {
      float d = 0.f;
      for (int s = 0; s < nsplit; ++s) d += work[((size_t)s * P + p) * PCF_SLOTS + k];
      const int64_t fo = (p / hw) * flag_set_stride;         // per-sample flags when every sample has its own prototypes
      w = expf(-sqrtf(d) / tau) * (k < C ? inst_flag[fo + k] : run_flag[fo + k - C]);
    }
    sw[px][k] = w;
  }
  for (int e = tid; e < PFU_PX * C; e += 256) {
    const int c = e / PFU_PX, px = e - c * PFU_PX;
    const int64_t p = p0 + px;
    if (p < P) {
      const int64_t b = p / hw;
      sl[px][c] = logits[((size_t)b * C + c) * hw + (p - b * hw)];
    }
  }
  __syncthreads();
  if (tid < PFU_PX && p0 + tid < P) {
    float si = 0.f, sr = 0.f, mx = -INFINITY;
    for (int c = 0; c < C; ++c) {
      si += sw[tid][c];
      sr += sw[tid][C + c];
      mx = fmaxf(mx, sl[tid][c]);
    }
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(sl[tid][c] - mx);
    sn[tid][0] = si, sn[tid][1] = sr, sn[tid][2] = mx, sn[tid][3] = se;
  }
  __syncthreads();
  for (int e = tid; e < PFU_PX * C; e += 256) {
    const int c = e / PFU_PX, px = e - c * PFU_PX;
    const int64_t p = p0 + px;
    if (p < P) {
      const int64_t b = p / hw;
      const float si = sn[px][0], sr = sn[px][1], mx = sn[px][2], se = sn[px][3];
      const float multi = pl * (sw[px][C + c] / sr) + pl1 * (sw[px][c] / si);     // ttda:1709 (0/0 -> NaN kept)
      out[((size_t)b * C + c) * hw + (p - b * hw)] = multi * fl + (expf(sl[px][c] - mx) / se) * fl1;   // ttda:1644
    }
  }
}

}  // namespace

extern "C" int diga_head_finish_pseudo_label(const float* partials, int splitk, int cpad, const float* bias_sum,
                                             const float* logits_in, float* logits_out, int32_t* labels, int B, int C, int hw,
                                             float threshold, int64_t label_sample_stride, void* stream) {
  DIGA_CHECK_ARG((partials != nullptr) != (logits_in != nullptr), "diga_head_finish_pseudo_label: give partials or logits_in");
  DIGA_CHECK_ARG(labels && B > 0 && hw > 0 && C > 0 && C <= DIGA_NUM_CLASSES_MAX, "diga_head_finish_pseudo_label: bad argument");
  DIGA_CHECK_ARG(!partials || (bias_sum && splitk >= 1 && cpad >= C), "diga_head_finish_pseudo_label: bad partials description");
  DIGA_CHECK_ARG(label_sample_stride == 0 || label_sample_stride >= hw, "diga_head_finish_pseudo_label: label stride < hw");
  const int64_t P = (int64_t)B * hw;
  hipLaunchKernelGGL(head_finish_pl_kernel, dim3(cdiv(P, 256)), dim3(256), 0, (hipStream_t)stream, partials, splitk, cpad,
                     bias_sum, logits_in, logits_out, labels, P, C, hw, threshold, label_sample_stride ? label_sample_stride : (int64_t)hw);
  DIGA_CHECK_LAUNCH("diga_head_finish_pseudo_label");
  return DIGA_OK;
}

extern "C" int diga_head_tapsum_pseudo_label(const float* z, int zstride, int ntaps, const int16_t* dy_host, const int16_t* dx_host,
                                             const float* bias_sum, float* logits_out, int32_t* labels, int B, int C, int h, int w,
                                             float threshold, int64_t label_sample_stride, void* stream) {
  DIGA_CHECK_ARG(z && dy_host && dx_host && bias_sum && logits_out && labels, "diga_head_tapsum_pseudo_label: NULL pointer");
  DIGA_CHECK_ARG(B > 0 && h > 0 && w > 0 && C > 0 && C <= 32 && ntaps >= 1 && ntaps <= DIGA_MAX_TAPS && zstride >= ntaps * C,
                 "diga_head_tapsum_pseudo_label: bad sizes (C <= 32, ntaps <= %d, zstride >= ntaps*C)", DIGA_MAX_TAPS);
  DIGA_CHECK_ARG(label_sample_stride == 0 || label_sample_stride >= (int64_t)h * w, "diga_head_tapsum_pseudo_label: label stride < h*w");
  TapList tl;
  tl.n = ntaps;
  for (int t = 0; t < DIGA_MAX_TAPS; ++t) {
    tl.dy[t] = t < ntaps ? dy_host[t] : 0;
    tl.dx[t] = t < ntaps ? dx_host[t] : 0;
  }
  const int64_t P = (int64_t)B * h * w;
  hipLaunchKernelGGL(head_tapsum_pl_kernel, dim3(cdiv(P, HTS_PX)), dim3(256), 0, (hipStream_t)stream, z, zstride, tl, bias_sum, logits_out,
                     labels, P, C, h, w, threshold, label_sample_stride ? label_sample_stride : (int64_t)h * w);
  DIGA_CHECK_LAUNCH("diga_head_tapsum_pseudo_label");
  return DIGA_OK;
}

extern "C" int diga_nearest_down_labels(const void* y, int y_is_i64, int32_t* out, int B, int H, int W, int h, int w,
                                        int64_t out_sample_stride, void* stream) {
  DIGA_CHECK_ARG(y && out && B > 0 && H > 0 && W > 0 && h > 0 && w > 0, "diga_nearest_down_labels: bad argument");
  DIGA_CHECK_ARG(out_sample_stride == 0 || out_sample_stride >= (int64_t)h * w, "diga_nearest_down_labels: output stride < h*w");
  const int64_t os = out_sample_stride ? out_sample_stride : (int64_t)h * w;
  const int64_t n = (int64_t)B * h * w;
  if (y_is_i64)
    hipLaunchKernelGGL(nearest_down_kernel<int64_t>, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const int64_t*)y, out,
                       B, H, W, h, w, os);
  else
    hipLaunchKernelGGL(nearest_down_kernel<int32_t>, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const int32_t*)y, out,
                       B, H, W, h, w, os);
  DIGA_CHECK_LAUNCH("diga_nearest_down_labels");
  return DIGA_OK;
}

namespace {
struct PoolPlan {
  int CS, JC;
  int64_t idx_bytes, partial_bytes;
};
PoolPlan pool_plan(int nsets, int64_t P, int C, int CH) {      // nsets = label sets in the call (frames x sets per frame)
  PoolPlan pp;
  // chunk = list positions one block sums: ~P/256 so that the dominant classes alone spread over the whole chip
  // (a chunk is a latency-bound chain of row reads; the partial buffer holds a set's chunks compactly, class after class:
  // at most JC + C chunks per set, i.e. nsets*(JC + C)*CH floats)
  pp.CS = (int)((P + 255) / 256);
  if (pp.CS < 32) pp.CS = 32;
  pp.JC = (int)((P + pp.CS - 1) / pp.CS);
  pp.idx_bytes = (((int64_t)nsets * C * P * 4) + 255) / 256 * 256;
  pp.partial_bytes = (int64_t)nsets * (pp.JC + C) * CH * 4;
  return pp;
}
}  // namespace

extern "C" int64_t diga_proto_pool_frames_work_bytes(int nframes, int nsets, int64_t P, int C, int CH) {
  if (nframes < 1 || nframes > 4096 || nsets < 1 || nsets > 2 || P <= 0 || C <= 0 || CH <= 0) return DIGA_EINVAL;
  const PoolPlan pp = pool_plan(nframes * nsets, P, C, CH);
  return pp.idx_bytes + pp.partial_bytes;
}

extern "C" int64_t diga_proto_pool_work_bytes(int nsets, int64_t P, int C, int CH) {
  return diga_proto_pool_frames_work_bytes(1, nsets, P, C, CH);
}

namespace {
int pool_frames(const void* feat, bool feat_f32, const int32_t* labels, int nframes, int nsets, int64_t P, int C, int CH, void* work,
                float* sums, int32_t* counts, void* stream) {
  DIGA_CHECK_ARG(feat && labels && work && sums && counts, "diga_proto_pool: NULL pointer");
  DIGA_CHECK_ARG(nframes >= 1 && nframes <= 4096 && nsets >= 1 && nsets <= 2 && P > 0 && P < (1ll << 31) && C > 0 &&
                     C <= DIGA_NUM_CLASSES_MAX && CH > 0 && CH % 8 == 0 && (int64_t)nframes * nsets * C <= 65535,
                 "diga_proto_pool: bad sizes");
  const int ns = nframes * nsets;
  const PoolPlan pp = pool_plan(ns, P, C, CH);
  int32_t* idx = (int32_t*)work;
  float* partial = (float*)((char*)work + pp.idx_bytes);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(proto_index_kernel, dim3(C, ns), dim3(64), 0, st, labels, P, C, idx, counts);
  if (feat_f32)
    hipLaunchKernelGGL(proto_pool_kernel<float>, dim3(pp.JC, C, ns), dim3(256), 0, st, (const float*)feat, idx, counts, P, C, CH,
                       pp.CS, pp.JC, partial, nsets);
  else
    hipLaunchKernelGGL(proto_pool_kernel<unsigned short>, dim3(pp.JC, C, ns), dim3(256), 0, st, (const unsigned short*)feat, idx,
                       counts, P, C, CH, pp.CS, pp.JC, partial, nsets);
  hipLaunchKernelGGL(proto_reduce_kernel, dim3(cdiv(CH, 256), ns * C), dim3(256), 0, st, partial, counts, C, CH, pp.CS, pp.JC,
                     sums);
  DIGA_CHECK_LAUNCH("diga_proto_pool");
  return DIGA_OK;
}
}  // namespace

extern "C" int diga_proto_pool_frames(const void* feat, const int32_t* labels, int nframes, int nsets, int64_t P, int C, int CH,
                                      void* work, float* sums, int32_t* counts, void* stream) {
  return pool_frames(feat, false, labels, nframes, nsets, P, C, CH, work, sums, counts, stream);
}

extern "C" int diga_proto_pool_f32(const float* feat, const int32_t* labels, int nsets, int64_t P, int C, int CH, void* work,
                                   float* sums, int32_t* counts, void* stream) {
  return pool_frames(feat, true, labels, 1, nsets, P, C, CH, work, sums, counts, stream);
}

extern "C" int diga_proto_pool(const void* feat, const int32_t* labels, int nsets, int64_t P, int C, int CH, void* work,
                               float* sums, int32_t* counts, void* stream) {
  return diga_proto_pool_frames(feat, labels, 1, nsets, P, C, CH, work, sums, counts, stream);
}

extern "C" int diga_proto_update(const float* sums, const int32_t* counts, int C, int CH, float* inst_proto, float* inst_flag,
                                 float* run_proto, float* run_flag, int32_t* state_flag, float one_minus_rho, float rho,
                                 void* stream) {
  DIGA_CHECK_ARG(sums && counts && inst_proto && inst_flag && run_proto && run_flag && state_flag, "diga_proto_update: NULL pointer");
  DIGA_CHECK_ARG(C > 0 && C <= DIGA_NUM_CLASSES_MAX && CH > 0, "diga_proto_update: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(proto_update_kernel, dim3(cdiv(CH, 256), C), dim3(256), 0, st, sums, counts, C, CH, inst_proto, inst_flag,
                     run_proto, run_flag, state_flag, one_minus_rho, rho);
  hipLaunchKernelGGL(proto_flags_kernel, dim3(1), dim3(64), 0, st, counts, C, inst_flag, run_flag, state_flag, rho);
  DIGA_CHECK_LAUNCH("diga_proto_update");
  return DIGA_OK;
}

extern "C" int diga_proto_update_frames(const float* sums, const int32_t* counts, int nframes, int C, int CH, float* inst_protos,
                                        float* inst_flags, float* run_protos, float* run_flags, float* run_proto, float* run_flag,
                                        int32_t* state_flag, float one_minus_rho, float rho, void* stream) {
  DIGA_CHECK_ARG(sums && counts && inst_protos && inst_flags && run_protos && run_flags && run_proto && run_flag && state_flag,
                 "diga_proto_update_frames: NULL pointer");
  DIGA_CHECK_ARG(nframes > 0 && C > 0 && C <= DIGA_NUM_CLASSES_MAX && CH > 0, "diga_proto_update_frames: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(proto_update_frames_kernel, dim3(cdiv(CH, 256), C), dim3(256), 0, st, sums, counts, nframes, C, CH, inst_protos,
                     run_protos, run_proto, run_flag, state_flag, one_minus_rho, rho);
  hipLaunchKernelGGL(proto_flags_frames_kernel, dim3(1), dim3(64), 0, st, counts, nframes, C, inst_flags, run_flags, run_flag,
                     state_flag, rho);
  DIGA_CHECK_LAUNCH("diga_proto_update_frames");
  return DIGA_OK;
}

extern "C" int64_t diga_proto_classify_frames_work_bytes(int nsets, int64_t P, int CH) {
  if (nsets <= 0 || P <= 0 || CH <= 0) return DIGA_EINVAL;
  return (int64_t)nsets * CH * 64 * 4 + (int64_t)PCF_SPLIT * P * PCF_SLOTS * 4;     // protoT per set + partial squared distances
}

extern "C" int64_t diga_proto_classify_work_bytes(int64_t P, int CH) { return diga_proto_classify_frames_work_bytes(1, P, CH); }

namespace {
int classify_fuse(const void* feat, const float* logits, const float* inst_proto, const float* inst_flag, const float* run_proto,
                  const float* run_flag, float* out, void* work, int B, int C, int hw, int CH, float tau, float pl, float pl1, float fl,
                  float fl1, int per_sample, void* stream, bool feat_f32 = false) {
  DIGA_CHECK_ARG(feat && logits && inst_proto && inst_flag && run_proto && run_flag && out && work, "diga_proto_classify_fuse: NULL pointer");
  DIGA_CHECK_ARG(B > 0 && hw > 0 && C > 0 && 2 * C <= PCF_SLOTS && CH > 0 && CH % 64 == 0,
                 "diga_proto_classify_fuse: bad sizes (C <= 20, CH %% 64 == 0)");
  const int64_t P = (int64_t)B * hw;
  int nsplit = PCF_SPLIT;                       // channel slices: the largest divisor of CH/64 that is <= PCF_SPLIT
  while ((CH / 64) % nsplit) --nsplit;
  hipStream_t st = (hipStream_t)stream;
  float* protoT = (float*)work;
  float* partial;
  const int nsets = per_sample ? B : 1;
  const int bps = cdiv(hw, PCF_PIX);
  DIGA_CHECK_ARG((int64_t)B * bps < (1ll << 31), "diga_proto_classify_fuse: too many pixels");
  partial = protoT + (size_t)nsets * CH * 64;
  hipLaunchKernelGGL(proto_transpose_kernel, dim3(cdiv(CH * 64, 256), nsets), dim3(256), 0, st, inst_proto, run_proto, protoT, C, CH);
  if (feat_f32)
    hipLaunchKernelGGL(proto_dist_kernel<float>, dim3(B * bps, nsplit), dim3(128), 0, st, (const float*)feat, protoT, partial, P, CH,
                       CH / nsplit, hw, bps, per_sample ? (int64_t)CH * 64 : 0);
  else
    hipLaunchKernelGGL(proto_dist_kernel<unsigned short>, dim3(B * bps, nsplit), dim3(128), 0, st, (const unsigned short*)feat, protoT,
                       partial, P, CH, CH / nsplit, hw, bps, per_sample ? (int64_t)CH * 64 : 0);
  hipLaunchKernelGGL(proto_fuse_kernel, dim3(cdiv(P, PFU_PX)), dim3(256), 0, st, (const float*)partial, logits, inst_flag, run_flag, out,
                     P, C, hw, nsplit, tau, pl, pl1, fl, fl1, per_sample ? C : 0);
  DIGA_CHECK_LAUNCH("diga_proto_classify_fuse");
  return DIGA_OK;
}
}  // namespace

extern "C" int diga_proto_classify_fuse(const void* feat, const float* logits, const float* inst_proto, const float* inst_flag,
                                        const float* run_proto, const float* run_flag, float* out, void* work, int B, int C, int hw,
                                        int CH, float tau, float pl, float pl1, float fl, float fl1, void* stream) {
  return classify_fuse(feat, logits, inst_proto, inst_flag, run_proto, run_flag, out, work, B, C, hw, CH, tau, pl, pl1, fl, fl1, 0, stream);
}

extern "C" int diga_proto_classify_fuse_f32(const float* feat, const float* logits, const float* inst_proto, const float* inst_flag,
                                            const float* run_proto, const float* run_flag, float* out, void* work, int B, int C, int hw,
                                            int CH, float tau, float pl, float pl1, float fl, float fl1, void* stream) {
  return classify_fuse(feat, logits, inst_proto, inst_flag, run_proto, run_flag, out, work, B, C, hw, CH, tau, pl, pl1, fl, fl1, 0, stream,
                       true);
}

extern "C" int diga_proto_classify_fuse_frames(const void* feat, const float* logits, const float* inst_protos, const float* inst_flags,
                                               const float* run_protos, const float* run_flags, float* out, void* work, int nframes,
                                               int C, int hw, int CH, float tau, float pl, float pl1, float fl, float fl1, void* stream) {
  return classify_fuse(feat, logits, inst_protos, inst_flags, run_protos, run_flags, out, work, nframes, C, hw, CH, tau, pl, pl1, fl, fl1,
                       1, stream);
}

// ---------------------------------------------------------------------------------------------
// The same classifier for callers that cannot allocate (code under HIP-graph capture, tight serving loops,
// harnesses that forbid hipMalloc): the work buffer of diga_proto_classify_fuse (prototype image + partial squared
// distances) lives in a file-scope device array instead of a caller-supplied workspace.  No counterpart in the
// reference.  Limits: B*hw <= DIGA_NOALLOC_MAX_PIXELS, CH <= DIGA_NOALLOC_MAX_CH.  Every device the library runs on
// holds its own copy of the array (20.5 MiB) until the process ends.
// ---------------------------------------------------------------------------------------------
constexpr int64_t NOALLOC_BYTES =
    (int64_t)DIGA_NOALLOC_MAX_CH * 64 * 4 + (int64_t)PCF_SPLIT * DIGA_NOALLOC_MAX_PIXELS * PCF_SLOTS * 4;
static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];

namespace {
constexpr int NOALLOC_MAX_DEVICES = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[NOALLOC_MAX_DEVICES];      // per device: the DEVICE address of that device's copy of `scratch`

// `scratch` in host code is only a handle for hipGetSymbolAddress, never a device address: resolve it once per
// device (with that device current) and hand the kernels the resolved pointer.
int scratch_for_current_device(void** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= NOALLOC_MAX_DEVICES) {
    diga_set_error("diga_proto_classify_fuse_noalloc: hipGetDevice failed or device index %d >= %d", dev, NOALLOC_MAX_DEVICES);
    return DIGA_ELAUNCH;
  }
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* ptr = nullptr;
    e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) {
      diga_set_error("diga_proto_classify_fuse_noalloc: hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e));
      return DIGA_ELAUNCH;
    }
    g_scratch_ptr[dev] = ptr;
  }
  *out = g_scratch_ptr[dev];
  return DIGA_OK;
}
}  // namespace

extern "C" int diga_proto_classify_fuse_noalloc(const void* feat, const float* logits, const float* inst_proto,
                                                const float* inst_flag, const float* run_proto, const float* run_flag,
                                                float* out, int B, int C, int hw, int CH, float tau, float pl, float pl1,
                                                float fl, float fl1, void* stream) {
  DIGA_CHECK_ARG(B > 0 && hw > 0 && CH > 0, "diga_proto_classify_fuse_noalloc: bad sizes");
  DIGA_CHECK_ARG((int64_t)B * hw <= DIGA_NOALLOC_MAX_PIXELS && CH <= DIGA_NOALLOC_MAX_CH,
                 "diga_proto_classify_fuse_noalloc: B*hw = %lld pixels x %d channels exceeds the static scratch (max %d pixels, %d "
                 "channels): use diga_proto_classify_fuse with a caller-supplied workspace",
                 (long long)B * hw, CH, DIGA_NOALLOC_MAX_PIXELS, DIGA_NOALLOC_MAX_CH);
  DIGA_CHECK_ARG(diga_proto_classify_frames_work_bytes(1, (int64_t)B * hw, CH) <= NOALLOC_BYTES,
                 "diga_proto_classify_fuse_noalloc: workspace plan exceeds the static scratch");
  void* ptr = nullptr;
  if (int e = scratch_for_current_device(&ptr)) return e;
  // nothing has to be cleared: the transpose kernel writes the whole prototype image, the distance kernel every partial
  // (all PCF_SLOTS slots of every pixel < P) the fuse kernel reads.  Calls that overlap on one GPU share the array: they race.
  return classify_fuse(feat, logits, inst_proto, inst_flag, run_proto, run_flag, out, ptr, B, C, hw, CH, tau, pl, pl1, fl, fl1, 0, stream);
}
