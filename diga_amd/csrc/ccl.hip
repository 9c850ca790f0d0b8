// Optional SAM extension (BASELINE.json north_star (2)): connected components of the per-class masks of a
// label map and instance-masked prototype pooling.  The reference (Waybaba/DIGA) has NO such code -- the default DIGA
// path never calls these kernels, and with respect to the reference their results are "parity unpinned"; the
// specification is the build's own and is pinned against scipy.ndimage.label (tests/golden/fn_ccl.npz).
//
// diga_ccl_label: union-find over pixels, parent = linear pixel index inside the sample.
//   init     parent[p] = the lowest-index same-class neighbour among NW, N, NE, W (a forest: pointers only decrease)
//   compress parent[p] = root(p)
//   merge    the scan-mask decision tree (N | NE+NW | NE+W | NE | NW | W): unite(p, q) with atomicMin on roots, so a
//            set's root is always its lowest pixel index = its first pixel in raster order
//   compress + count   roots per (256-pixel block, class) by wave ballots               -> blockcnt, rank inside block
//   scan     per (sample, class) exclusive scan over the blocks                         -> blockbase, ncomp
//   gather   inst[p] = blockbase[block of root][class] + rank[root] + 1   (raster order of first pixels, per class)
//   (A one-launch variant -- one block per sample, parent array in LDS -- was measured SLOWER than these six launches:
//   57 us against 28 us for fifteen 65x129 frames, one CU walking its frame in nine serial strides; it is not built.)
// diga_instance_pool: stable per-instance pixel lists (one wave per instance walks [first, last] of its pixels),
//   chunked register-accumulated row sums, fixed-order reduce -> deterministic means, no float atomics.
#include "common.h"

namespace {

constexpr int CCL_BLK = 256;      // pixels per block of the global path
constexpr int IP_CS = 64;         // list positions per pooling job

__device__ __forceinline__ int cls_of(int v, int C) { return (unsigned)v < (unsigned)C ? v : -1; }

// ---- union-find on a parent array in global memory.  Loads and atomics are agent-scope: the blocks of one launch run
// on different XCDs, whose L2s are not coherent for plain accesses.  A stale parent is still an ancestor (pointers only
// ever move towards lower indices inside the same set), so readers never need more than that.
__device__ __forceinline__ int g_load(const int* L, int i) { return __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int g_find(const int* L, int i) {
  int p;
  while ((p = g_load(L, i)) != i) i = p;
  return i;
}
__device__ __forceinline__ void g_unite(int* L, int a, int b) {
  for (;;) {
    a = g_find(L, a);
    b = g_find(L, b);
    if (a == b) return;
    if (a > b) {
      const int t = a;
      a = b;
      b = t;
    }
    const int old = __hip_atomic_fetch_min(L + b, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // hang root b under a
    if (old == b) return;
    b = old;                        // b was re-parented meanwhile: carry on from its new parent
  }
}
// same-class flags of the four scan-mask neighbours of pixel (x, y) of class c
struct Nbr {
  bool w, n, nw, ne;
};
__device__ __forceinline__ Nbr neighbours(const int32_t* __restrict__ lab, int p, int x, int y, int w, int c, bool conn8) {
  Nbr r;
  r.w = x > 0 && lab[p - 1] == c;
  r.n = y > 0 && lab[p - w] == c;
  r.nw = conn8 && y > 0 && x > 0 && lab[p - w - 1] == c;
  r.ne = conn8 && y > 0 && x + 1 < w && lab[p - w + 1] == c;
  return r;
}
__device__ __forceinline__ int first_parent(const Nbr& r, int p, int w) {
  return r.nw ? p - w - 1 : r.n ? p - w : r.ne ? p - w + 1 : r.w ? p - 1 : p;
}
// The unions a pixel owes (8-connectivity: the classic scan-mask decision tree -- N alone suffices when present, because
// NW, NE and W are then tied to N by their own rows; 4-connectivity: N and W).  f(q) is called for each partner q.
template <typename F>
__device__ __forceinline__ void owed_unions(const Nbr& r, int p, int w, bool conn8, F&& f) {
  if (!conn8) {
    if (r.n) f(p - w);
    if (r.w) f(p - 1);
    return;
  }
  if (r.n) {
    f(p - w);
  } else if (r.ne) {
    f(p - w + 1);
    if (r.nw) f(p - w - 1);
    else if (r.w) f(p - 1);
  } else if (r.nw) {
    f(p - w - 1);
  } else if (r.w) {
    f(p - 1);
  }
}

__global__ __launch_bounds__(CCL_BLK) void ccl_init_kernel(const int32_t* __restrict__ labels, int64_t lstride, int h, int w, int C,
                                                           int conn8, int* __restrict__ parent) {
  const int hw = h * w, p = blockIdx.x * CCL_BLK + threadIdx.x;
  if (p >= hw) return;
  const int32_t* lab = labels + blockIdx.y * lstride;
  int* L = parent + (size_t)blockIdx.y * hw;
  const int c = cls_of(lab[p], C);
  if (c < 0) {
    L[p] = -1;
    return;
  }
  const int y = p / w, x = p - y * w;
  L[p] = first_parent(neighbours(lab, p, x, y, w, c, conn8), p, w);
}

// Root by plain (L2-cached) loads, for the launches in which nobody unites: the parents were written by an EARLIER launch
// (visible after the kernel boundary) or are being overwritten with roots by this one -- either value is an ancestor.
__device__ __forceinline__ int q_find(const int* L, int i) {
  int p;
  while ((p = L[i]) != i) i = p;
  return i;
}

__global__ __launch_bounds__(CCL_BLK) void ccl_compress_kernel(int hw, int* parent) {
  const int p = blockIdx.x * CCL_BLK + threadIdx.x;
  if (p >= hw) return;
  int* L = parent + (size_t)blockIdx.y * hw;
  const int v = L[p];
  if (v >= 0 && v != p) L[p] = q_find(L, v);
}

__global__ __launch_bounds__(CCL_BLK) void ccl_merge_kernel(const int32_t* __restrict__ labels, int64_t lstride, int h, int w, int C,
                                                            int conn8, int* parent) {
  const int hw = h * w, p = blockIdx.x * CCL_BLK + threadIdx.x;
  if (p >= hw) return;
  const int32_t* lab = labels + blockIdx.y * lstride;
  int* L = parent + (size_t)blockIdx.y * hw;
  const int c = cls_of(lab[p], C);
  if (c < 0) return;
  const int y = p / w, x = p - y * w;
  // After the compress launch nearly every pixel already shares its root with the neighbours it owes a union: one plain
  // (L2-cached) look at both parents settles those.  The values may be stale by now, but "equal => same set" stays true
  // because sets only ever merge.  Only the rest pays for agent-scope finds and atomics.
  const int rp = L[p];
  owed_unions(neighbours(lab, p, x, y, w, c, conn8), p, w, conn8, [&](int q) {
    if (L[q] != rp) g_unite(L, p, q);
  });
}

// Rank of a root pixel among the roots of its class inside a group of NWV waves (raster order), and the group's count
// per class.  Every thread of the block must call this.  wcnt: LDS [NWV][DIGA_NUM_CLASSES_MAX].
template <int NWV>
__device__ __forceinline__ int class_rank(bool is_root, int c, int C, int (*wcnt)[DIGA_NUM_CLASSES_MAX]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  int rank = 0;
  for (int k = 0; k < C; ++k) {
    const unsigned long long m = __ballot(is_root && c == k);
    if (c == k) rank = __popcll(m & below);
    if (lane == 0) wcnt[wave][k] = __popcll(m);
  }
  __syncthreads();
  if (is_root)
    for (int v = 0; v < wave; ++v) rank += wcnt[v][c];
  return rank;
}

__global__ __launch_bounds__(CCL_BLK) void ccl_compress_count_kernel(const int32_t* __restrict__ labels, int64_t lstride, int hw, int C,
                                                                     int nblk, int* parent, int* __restrict__ lrank,
                                                                     int* __restrict__ blockcnt) {
  __shared__ int wcnt[CCL_BLK / 64][DIGA_NUM_CLASSES_MAX];
  const int p = blockIdx.x * CCL_BLK + threadIdx.x;
  const int32_t* lab = labels + blockIdx.y * lstride;
  int* L = parent + (size_t)blockIdx.y * hw;
  int c = -1;
  bool is_root = false;
  if (p < hw) {
    c = cls_of(lab[p], C);
    if (c >= 0) {
      const int r = q_find(L, p);
      L[p] = r;
      is_root = r == p;
    }
  }
  const int rank = class_rank<CCL_BLK / 64>(is_root, c, C, wcnt);
  if (is_root) lrank[(size_t)blockIdx.y * hw + p] = rank;
  if (threadIdx.x < C) {
    int t = 0;
#pragma unroll
    for (int v = 0; v < CCL_BLK / 64; ++v) t += wcnt[v][threadIdx.x];
    blockcnt[((size_t)blockIdx.y * nblk + blockIdx.x) * C + threadIdx.x] = t;
  }
}

__device__ __forceinline__ int wave_inclusive_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}

// one block per (class, sample): exclusive scan of the per-block root counts over the blocks.  Thread t owns the
// contiguous segment [t*seg, (t+1)*seg) of blocks: segment sums, a block-wide scan of those, then the running prefix.
__global__ __launch_bounds__(256) void ccl_scan_kernel(const int* __restrict__ blockcnt, int nblk, int C, int* __restrict__ blockbase,
                                                       int32_t* __restrict__ ncomp) {
  __shared__ int wsum[4];
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int seg = (nblk + 255) / 256;
  const int i0 = min(tid * seg, nblk), i1 = min(i0 + seg, nblk);
  const int* src = blockcnt + (size_t)b * nblk * C + c;
  int* dst = blockbase + (size_t)b * nblk * C + c;
  int mine = 0;
  for (int i = i0; i < i1; ++i) mine += src[(size_t)i * C];
  const int inc = wave_inclusive_scan(mine);
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int run = inc - mine;
  for (int v = 0; v < wave; ++v) run += wsum[v];
  for (int i = i0; i < i1; ++i) {
    dst[(size_t)i * C] = run;
    run += src[(size_t)i * C];
  }
  if (tid == 255) ncomp[b * C + c] = run;
}

__global__ __launch_bounds__(CCL_BLK) void ccl_gather_kernel(const int32_t* __restrict__ labels, int64_t lstride, int hw, int C,
                                                             int nblk, const int* __restrict__ parent, const int* __restrict__ lrank,
                                                             const int* __restrict__ blockbase, int32_t* __restrict__ inst) {
  const int p = blockIdx.x * CCL_BLK + threadIdx.x;
  if (p >= hw) return;
  const int b = blockIdx.y;
  const int c = cls_of(labels[b * lstride + p], C);
  int v = 0;
  if (c >= 0) {
    const int r = parent[(size_t)b * hw + p];
    v = blockbase[((size_t)b * nblk + r / CCL_BLK) * C + c] + lrank[(size_t)b * hw + r] + 1;
  }
  inst[(size_t)b * hw + p] = v;
}

struct CclPlan {
  int nblk;
  int64_t parent_bytes, cnt_bytes;
};
CclPlan ccl_plan(int B, int h, int w, int C) {
  CclPlan pl;
  pl.nblk = cdiv((int64_t)h * w, CCL_BLK);
  pl.parent_bytes = ((int64_t)B * h * w * 4 + 255) / 256 * 256;
  pl.cnt_bytes = ((int64_t)B * pl.nblk * C * 4 + 255) / 256 * 256;
  return pl;
}

// ------------------------------------------------------------------------------------------------------------------
// instance pooling
// ------------------------------------------------------------------------------------------------------------------
// block 0 / wave 0: base[b*C + c] = exclusive scan of ncomp (rows of the instance table are ordered sample, class,
// component number), base[B*C] = total; all blocks: reset the per-instance counters
__global__ __launch_bounds__(256) void inst_init_kernel(const int32_t* __restrict__ ncomp, int BC, int max_inst, int* __restrict__ base,
                                                        int* __restrict__ cnt, int* __restrict__ first, int* __restrict__ last,
                                                        int32_t* __restrict__ n_total) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g < max_inst) {
    cnt[g] = 0;
    first[g] = 0x7fffffff;
    last[g] = -1;
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    const int lane = threadIdx.x;
    int run = 0;
    for (int i0 = 0; i0 < BC; i0 += 64) {
      const int i = i0 + lane;
      const int v = i < BC ? ncomp[i] : 0;
      const int inc = wave_inclusive_scan(v);
      if (i < BC) base[i] = run + inc - v;
      run += __shfl(inc, 63);
    }
    if (lane == 0) {
      base[BC] = run;
      *n_total = run;
    }
  }
}

__global__ __launch_bounds__(256) void inst_key_kernel(const int32_t* __restrict__ labels, int64_t lstride, const int32_t* __restrict__ inst,
                                                       const int* __restrict__ base, int64_t hw, int64_t P, int C, int max_inst,
                                                       int* __restrict__ key, int* __restrict__ cnt, int* __restrict__ first,
                                                       int* __restrict__ last) {
  const int64_t pg = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int g = -1;
  if (pg < P) {
    const int b = (int)(pg / hw);
    const int64_t p = pg - (int64_t)b * hw;
    const int c = cls_of(labels[b * lstride + p], C);
    const int i = inst[pg];
    if (c >= 0 && i > 0) {
      g = base[b * C + c] + i - 1;
      if (g >= max_inst) g = -1;          // beyond the caller's table: dropped (n_total tells)
    }
    key[pg] = g;
  }
  // One set of atomics per (wave, instance) instead of per pixel: neighbouring pixels mostly share their instance, and a
  // large component would otherwise serialise thousands of updates of the same three words.  Lanes are in pixel order,
  // so the lowest / highest lane of a group holds its first / last pixel.  Integer atomics: order does not matter.
  unsigned long long todo = __ballot(g >= 0);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int g0 = __shfl(g, leader);
    const unsigned long long m = __ballot(g == g0);
    if (lane == leader) {
      const int lo = (int)(pg - lane) + leader, hi = (int)(pg - lane) + 63 - __clzll((long long)m);
      atomicAdd(cnt + g0, __popcll(m));
      atomicMin(first + g0, lo);
      atomicMax(last + g0, hi);
    }
    todo &= ~m;
  }
}

// single block: offs = exclusive scan of cnt, joffs = exclusive scan of the job counts ceil(cnt / IP_CS)
__global__ __launch_bounds__(1024) void inst_scan_kernel(const int* __restrict__ base, int BC, int max_inst, const int* __restrict__ cnt,
                                                         int* __restrict__ offs, int* __restrict__ joffs) {
  __shared__ int wsum[2][16];
  __shared__ int run[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = min(base[BC], max_inst);
  if (tid < 2) run[tid] = 0;
  __syncthreads();
  for (int g0 = 0; g0 < n; g0 += 1024) {
    const int g = g0 + tid;
    const int v = g < n ? cnt[g] : 0;
    const int jv = (v + IP_CS - 1) / IP_CS;
    const int inc = wave_inclusive_scan(v), jinc = wave_inclusive_scan(jv);
    if (lane == 63) {
      wsum[0][wave] = inc;
      wsum[1][wave] = jinc;
    }
    __syncthreads();
    int pre = run[0], jpre = run[1];
    for (int u = 0; u < wave; ++u) {
      pre += wsum[0][u];
      jpre += wsum[1][u];
    }
    if (g < n) {
      offs[g] = pre + inc - v;
      joffs[g] = jpre + jinc - jv;
    }
    __syncthreads();
    if (tid == 1023) {
      run[0] = pre + inc;
      run[1] = jpre + jinc;
    }
    __syncthreads();
  }
  if (tid == 0) {
    offs[n] = run[0];
    joffs[n] = run[1];
  }
}

// one wave per instance: the stable (pixel-order) list of its pixels, found by walking [first, last]
__global__ __launch_bounds__(64) void inst_list_kernel(const int* __restrict__ base, int BC, int max_inst, const int* __restrict__ key,
                                                       const int* __restrict__ first, const int* __restrict__ last,
                                                       const int* __restrict__ offs, int* __restrict__ list) {
  const int g = blockIdx.x, lane = threadIdx.x;
  if (g >= min(base[BC], max_inst)) return;
  const int p1 = last[g];
  int pos = offs[g];
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int p0 = first[g]; p0 <= p1; p0 += 64) {
    const int p = p0 + lane;
    const bool mine = p <= p1 && key[p] == g;
    const unsigned long long m = __ballot(mine);
    if (mine) list[pos + __popcll(m & below)] = p;
    pos += __popcll(m);
  }
}

// job j = chunk (j - joffs[g]) of instance g: sums IP_CS listed feature rows in registers -> partial[j][CH]
__global__ __launch_bounds__(256) void inst_pool_kernel(const unsigned short* __restrict__ feat, const int* __restrict__ base, int BC,
                                                        int max_inst, const int* __restrict__ cnt, const int* __restrict__ offs,
                                                        const int* __restrict__ joffs, const int* __restrict__ list, int CH,
                                                        float* __restrict__ partial, float* __restrict__ protos) {
  const int j = blockIdx.x;
  const int n = min(base[BC], max_inst);
  if (j >= joffs[n]) return;
  int lo = 0, hi = n - 1;                 // largest g with joffs[g] <= j (instances without pixels do not occur)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (joffs[mid] <= j) lo = mid;
    else hi = mid - 1;
  }
  const int g = lo;
  const int beg = offs[g] + (j - joffs[g]) * IP_CS;
  const int nc = cnt[g];
  const int end = min(beg + IP_CS, offs[g] + nc);
  // an instance of a single job (most are small) is finished here: its mean goes straight to the table
  const bool whole = joffs[g + 1] - joffs[g] == 1;
  float* dst = whole ? protos + (size_t)g * CH : partial + (size_t)j * CH;
  const float div = whole ? (float)nc : 1.f;
  for (int ch0 = threadIdx.x * 8; ch0 < CH; ch0 += 2048) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    int q = beg;
    for (; q + 8 <= end; q += 8) {
      u32x4 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = *reinterpret_cast<const u32x4*>(feat + (size_t)list[q + u] * CH + ch0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float f[8];
        unpack_bf16x8(r[u], f);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += f[i];
      }
    }
    for (; q < end; ++q) {
      float f[8];
      unpack_bf16x8(*reinterpret_cast<const u32x4*>(feat + (size_t)list[q] * CH + ch0), f);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += f[i];
    }
    *reinterpret_cast<f32x4*>(dst + ch0) = f32x4{acc[0] / div, acc[1] / div, acc[2] / div, acc[3] / div};
    *reinterpret_cast<f32x4*>(dst + ch0 + 4) = f32x4{acc[4] / div, acc[5] / div, acc[6] / div, acc[7] / div};
  }
}

// one block per table row: the row's meta data, and for instances of several pooling jobs the fixed-order sum of their partials
__global__ __launch_bounds__(256) void inst_reduce_kernel(const float* __restrict__ partial, const int* __restrict__ base, int BC, int C,
                                                          int max_inst, const int* __restrict__ cnt, const int* __restrict__ joffs,
                                                          int CH, float* __restrict__ protos, int32_t* __restrict__ counts,
                                                          int32_t* __restrict__ inst_class, int32_t* __restrict__ inst_sample) {
  const int g = blockIdx.x;
  const int n = min(base[BC], max_inst);
  if (g >= n) {
    if (threadIdx.x == 0) {
      counts[g] = 0;
      inst_class[g] = -1;
      inst_sample[g] = -1;
    }
    return;
  }
  const int nc = cnt[g];
  if (threadIdx.x == 0) {
    int lo = 0, hi = BC - 1;              // largest (b, c) with base <= g: the one whose components include row g
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (base[mid] <= g) lo = mid;
      else hi = mid - 1;
    }
    counts[g] = nc;
    inst_class[g] = lo % C;
    inst_sample[g] = lo / C;
  }
  const int j0 = joffs[g], j1 = joffs[g + 1];
  if (j1 - j0 == 1) return;               // written by its only pooling job
  for (int ch = threadIdx.x; ch < CH; ch += 256) {
    float v = 0.f;
    for (int j = j0; j < j1; ++j) v += partial[(size_t)j * CH + ch];
    protos[(size_t)g * CH + ch] = v / (float)nc;
  }
}

struct InstPlan {
  int max_jobs;
  int64_t o_key, o_list, o_cnt, o_first, o_last, o_offs, o_joffs, o_base, o_partial, total;
};
InstPlan inst_plan(int B, int64_t hw, int C, int CH, int max_inst) {
  InstPlan pl;
  const int64_t P = (int64_t)B * hw;
  auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
  pl.max_jobs = (int)((P + IP_CS - 1) / IP_CS) + max_inst;
  int64_t o = 0;
  pl.o_key = o, o += al(P * 4);
  pl.o_list = o, o += al(P * 4);
  pl.o_cnt = o, o += al((int64_t)max_inst * 4);
  pl.o_first = o, o += al((int64_t)max_inst * 4);
  pl.o_last = o, o += al((int64_t)max_inst * 4);
  pl.o_offs = o, o += al(((int64_t)max_inst + 1) * 4);
  pl.o_joffs = o, o += al(((int64_t)max_inst + 1) * 4);
  pl.o_base = o, o += al(((int64_t)B * C + 1) * 4);
  pl.o_partial = o, o += al((int64_t)pl.max_jobs * CH * 4);
  pl.total = o;
  return pl;
}

}  // namespace

extern "C" int64_t diga_ccl_work_bytes(int B, int h, int w, int C) {
  if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || C > DIGA_NUM_CLASSES_MAX || (int64_t)h * w >= (1ll << 31)) return DIGA_EINVAL;
  const CclPlan pl = ccl_plan(B, h, w, C);
  return 2 * pl.parent_bytes + 2 * pl.cnt_bytes;
}

extern "C" int diga_ccl_label(const int32_t* labels, int64_t label_sample_stride, int B, int h, int w, int C, int connectivity,
                              int32_t* inst, int32_t* ncomp, void* work, void* stream) {
  DIGA_CHECK_ARG(labels && inst && ncomp && work, "diga_ccl_label: NULL pointer");
  DIGA_CHECK_ARG(B > 0 && h > 0 && w > 0 && (int64_t)h * w < (1ll << 31) && (int64_t)B * h * w < (1ll << 40), "diga_ccl_label: bad sizes");
  DIGA_CHECK_ARG(C > 0 && C <= DIGA_NUM_CLASSES_MAX, "diga_ccl_label: C=%d outside 1..%d", C, DIGA_NUM_CLASSES_MAX);
  DIGA_CHECK_ARG(connectivity == 4 || connectivity == 8, "diga_ccl_label: connectivity %d (4 or 8)", connectivity);
  const int hw = h * w;
  const int64_t ls = label_sample_stride ? label_sample_stride : hw;
  DIGA_CHECK_ARG(ls >= hw, "diga_ccl_label: label_sample_stride < h*w");
  const int conn8 = connectivity == 8;
  hipStream_t st = (hipStream_t)stream;
  const CclPlan pl = ccl_plan(B, h, w, C);
  int* parent = (int*)work;
  int* lrank = (int*)((char*)work + pl.parent_bytes);
  int* blockcnt = (int*)((char*)work + 2 * pl.parent_bytes);
  int* blockbase = (int*)((char*)work + 2 * pl.parent_bytes + pl.cnt_bytes);
  const dim3 grid(pl.nblk, B), block(CCL_BLK);
  hipLaunchKernelGGL(ccl_init_kernel, grid, block, 0, st, labels, ls, h, w, C, conn8, parent);
  hipLaunchKernelGGL(ccl_compress_kernel, grid, block, 0, st, hw, parent);
  hipLaunchKernelGGL(ccl_merge_kernel, grid, block, 0, st, labels, ls, h, w, C, conn8, parent);
  hipLaunchKernelGGL(ccl_compress_count_kernel, grid, block, 0, st, labels, ls, hw, C, pl.nblk, parent, lrank, blockcnt);
  hipLaunchKernelGGL(ccl_scan_kernel, dim3(C, B), dim3(256), 0, st, blockcnt, pl.nblk, C, blockbase, ncomp);
  hipLaunchKernelGGL(ccl_gather_kernel, grid, block, 0, st, labels, ls, hw, C, pl.nblk, parent, lrank, blockbase, inst);
  DIGA_CHECK_LAUNCH("diga_ccl_label");
  return DIGA_OK;
}

extern "C" int64_t diga_instance_pool_work_bytes(int B, int64_t hw, int C, int CH, int max_inst) {
  if (B <= 0 || hw <= 0 || C <= 0 || C > DIGA_NUM_CLASSES_MAX || CH <= 0 || max_inst <= 0 || (int64_t)B * hw >= (1ll << 31))
    return DIGA_EINVAL;
  return inst_plan(B, hw, C, CH, max_inst).total;
}

extern "C" int diga_instance_pool(const void* feat, const int32_t* labels, int64_t label_sample_stride, const int32_t* inst,
                                  const int32_t* ncomp, int B, int64_t hw, int C, int CH, int max_inst, float* protos,
                                  int32_t* counts, int32_t* inst_class, int32_t* inst_sample, int32_t* n_total, void* work,
                                  void* stream) {
  DIGA_CHECK_ARG(feat && labels && inst && ncomp && protos && counts && inst_class && inst_sample && n_total && work,
                 "diga_instance_pool: NULL pointer");
  DIGA_CHECK_ARG(B > 0 && hw > 0 && (int64_t)B * hw < (1ll << 31) && C > 0 && C <= DIGA_NUM_CLASSES_MAX && CH > 0 && CH % 8 == 0 &&
                     max_inst > 0 && max_inst <= (1 << 20),
                 "diga_instance_pool: bad sizes");
  const int64_t ls = label_sample_stride ? label_sample_stride : hw;
  DIGA_CHECK_ARG(ls >= hw, "diga_instance_pool: label_sample_stride < hw");
  const InstPlan pl = inst_plan(B, hw, C, CH, max_inst);
  char* wk = (char*)work;
  int *key = (int*)(wk + pl.o_key), *list = (int*)(wk + pl.o_list), *cnt = (int*)(wk + pl.o_cnt), *first = (int*)(wk + pl.o_first),
      *last = (int*)(wk + pl.o_last), *offs = (int*)(wk + pl.o_offs), *joffs = (int*)(wk + pl.o_joffs), *base = (int*)(wk + pl.o_base);
  float* partial = (float*)(wk + pl.o_partial);
  const int64_t P = (int64_t)B * hw;
  const int BC = B * C;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(inst_init_kernel, dim3(cdiv(max_inst, 256)), dim3(256), 0, st, ncomp, BC, max_inst, base, cnt, first, last, n_total);
  hipLaunchKernelGGL(inst_key_kernel, dim3(cdiv(P, 256)), dim3(256), 0, st, labels, ls, inst, base, hw, P, C, max_inst, key, cnt, first,
                     last);
  hipLaunchKernelGGL(inst_scan_kernel, dim3(1), dim3(1024), 0, st, base, BC, max_inst, cnt, offs, joffs);
  hipLaunchKernelGGL(inst_list_kernel, dim3(max_inst), dim3(64), 0, st, base, BC, max_inst, key, first, last, offs, list);
  hipLaunchKernelGGL(inst_pool_kernel, dim3(pl.max_jobs), dim3(256), 0, st, (const unsigned short*)feat, base, BC, max_inst, cnt, offs,
                     joffs, list, CH, partial, protos);
  hipLaunchKernelGGL(inst_reduce_kernel, dim3(max_inst), dim3(256), 0, st, partial, base, BC, C, max_inst, cnt, joffs, CH,
                     protos, counts, inst_class, inst_sample);
  DIGA_CHECK_LAUNCH("diga_instance_pool");
  return DIGA_OK;
}
