// One-shot all-reduce over direct peer writes: the sharded path's exchanges are tiny and
// latency-bound -- per BN layer [2 groups][2][C] fp64 sums (1..64 KB), per step the prototype sums (311 KB) -- so a ring is
// the wrong algorithm.  Every rank WRITES its contribution straight into a slot of every peer's buffer (one hop over the
// direct xGMI link, peers mapped with hipIpc), raises a flag there, waits for the peers' flags on its own buffer and sums the
// slots in fixed rank order: one kernel launch per exchange, deterministic, identical bits on every rank, capturable in a
// HIP graph (the sequence number lives in device memory).  No counterpart in the reference (it has no distributed code).
//
// Memory model: the buffer is ordinary device memory shared through hipIpc.  Data and flags are written with system-scope
// stores / atomics and a system-scope fence in between, and read with system-scope atomic loads, so that neither the writer's
// nor the reader's L2 can hold them back or serve stale lines.  Waiting is a BOUNDED spin: a peer that never arrives makes the
// call fail (error word, DIGA_ELAUNCH from diga_xgmi_check) instead of hanging the GPU.
#include "common.h"
#include <string.h>

namespace {

constexpr int XG_MAX_WORLD = 8;
constexpr int XG_MAX_BLOCKS = 512;            // blocks of one launch (each owns XG_EPB elements and its own flag / epoch)
constexpr int XG_EPB = 2048;                  // 8-byte elements per block
constexpr unsigned XG_SPIN_LIMIT = 1u << 22;  // polls of ~1 us each (sleep + a system-scope load): a few seconds, then give up

struct XgDev {                                // passed to the kernel by value
  unsigned char* peer[XG_MAX_WORLD];          // every rank's buffer as mapped in THIS process (peer[rank] = the local one)
  int world, rank;
  long long max_elems;                        // 8-byte elements per slot
};

// buffer layout (bytes): [2 parities][world][max_elems] x 8 | flags [world][XG_MAX_BLOCKS] u32 | epoch [XG_MAX_BLOCKS] u32 | error u32
__host__ __device__ inline size_t xg_slot_off(const XgDev& d, int parity, int r) { return ((size_t)parity * d.world + r) * d.max_elems * 8; }
__host__ __device__ inline size_t xg_flags_off(const XgDev& d) { return (size_t)2 * d.world * d.max_elems * 8; }
__host__ __device__ inline size_t xg_epoch_off(const XgDev& d) { return xg_flags_off(d) + (size_t)d.world * XG_MAX_BLOCKS * 4; }
__host__ __device__ inline size_t xg_error_off(const XgDev& d) { return xg_epoch_off(d) + (size_t)XG_MAX_BLOCKS * 4; }
__host__ __device__ inline size_t xg_total_bytes(const XgDev& d) { return (xg_error_off(d) + 4 + 255) / 256 * 256; }

template <typename T>
struct Acc;
template <>
struct Acc<double> {
  static __device__ double from_bits(unsigned long long b) { return __longlong_as_double((long long)b); }
  static __device__ unsigned long long to_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
};
// float / int32 payloads travel as pairs packed into the 8-byte slot elements (element i = values 2i, 2i+1)

// dtype: 0 = fp64 (n elements), 1 = fp32 pairs, 2 = int32 pairs (n = number of 8-byte elements)
template <int DT>
__global__ __launch_bounds__(256) void xgmi_allreduce_kernel(const XgDev d, unsigned long long* __restrict__ data, long long n) {
  __shared__ int s_fail;
  const int tid = threadIdx.x, b = blockIdx.x;
  const long long lo = (long long)b * XG_EPB, hi = min(n, lo + XG_EPB);
  unsigned char* const me = d.peer[d.rank];
  unsigned* const epoch = reinterpret_cast<unsigned*>(me + xg_epoch_off(d));
  const unsigned e = epoch[b] + 1u;                       // this block's sequence number: the same on every rank
  const int parity = (int)(e & 1u);
  if (tid == 0) s_fail = 0;
  // 1. my contribution into slot [parity][rank] of every rank's buffer (mine included)
  for (int r = 0; r < d.world; ++r) {
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(d.peer[r] + xg_slot_off(d, parity, d.rank));
    for (long long i = lo + tid; i < hi; i += 256)
      __hip_atomic_store(dst + i, data[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  __syncthreads();
  if (tid < d.world) {
    unsigned* f = reinterpret_cast<unsigned*>(d.peer[tid] + xg_flags_off(d)) + (size_t)d.rank * XG_MAX_BLOCKS + b;
    __hip_atomic_store(f, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // 2. wait until every rank's contribution to this block has landed in MY buffer (bounded)
    const unsigned* mine = reinterpret_cast<const unsigned*>(me + xg_flags_off(d)) + (size_t)tid * XG_MAX_BLOCKS + b;
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > XG_SPIN_LIMIT) {
        s_fail = 1;
        break;
      }
    }
  }
  __syncthreads();
  if (s_fail) {                                           // a peer never arrived: report, leave data untouched, do not hang
    if (tid == 0) __hip_atomic_store(reinterpret_cast<unsigned*>(me + xg_error_off(d)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {
    // 3. sum the slots in rank order (the same order on every rank: identical bits everywhere)
    for (long long i = lo + tid; i < hi; i += 256) {
      unsigned long long out;
      if constexpr (DT == 0) {
        double v = 0.0;
        for (int r = 0; r < d.world; ++r)
          v += Acc<double>::from_bits(__hip_atomic_load(reinterpret_cast<const unsigned long long*>(me + xg_slot_off(d, parity, r)) + i,
                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        out = Acc<double>::to_bits(v);
      } else if constexpr (DT == 1) {
        float va = 0.f, vb = 0.f;
        for (int r = 0; r < d.world; ++r) {
          const unsigned long long q = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(me + xg_slot_off(d, parity, r)) + i,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          va += __uint_as_float((unsigned)(q & 0xffffffffull));
          vb += __uint_as_float((unsigned)(q >> 32));
        }
        out = (unsigned long long)__float_as_uint(va) | ((unsigned long long)__float_as_uint(vb) << 32);
      } else {
        int va = 0, vb = 0;
        for (int r = 0; r < d.world; ++r) {
          const unsigned long long q = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(me + xg_slot_off(d, parity, r)) + i,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          va += (int)(unsigned)(q & 0xffffffffull);
          vb += (int)(unsigned)(q >> 32);
        }
        out = (unsigned long long)(unsigned)va | ((unsigned long long)(unsigned)vb << 32);
      }
      data[i] = out;
    }
  }
  __syncthreads();
  if (tid == 0) epoch[b] = e;
}

struct XgComm {
  XgDev dev;
  bool connected = false;
  void* opened[XG_MAX_WORLD] = {nullptr};
};

}  // namespace

extern "C" int64_t diga_xgmi_buffer_bytes(int world, int64_t max_elems) {
  if (world < 1 || world > XG_MAX_WORLD || max_elems < 1 || max_elems > (int64_t)XG_MAX_BLOCKS * XG_EPB) return DIGA_EINVAL;
  XgDev d{};
  d.world = world, d.max_elems = max_elems;
  return (int64_t)xg_total_bytes(d);
}

extern "C" int diga_xgmi_create(int world, int rank, int64_t max_elems, void** comm_out, void* ipc_handle_out64) {
  DIGA_CHECK_ARG(world >= 1 && world <= XG_MAX_WORLD && rank >= 0 && rank < world, "diga_xgmi_create: world %d / rank %d", world, rank);
  DIGA_CHECK_ARG(max_elems >= 1 && max_elems <= (int64_t)XG_MAX_BLOCKS * XG_EPB, "diga_xgmi_create: max_elems %lld (limit %d)",
                 (long long)max_elems, XG_MAX_BLOCKS * XG_EPB);
  DIGA_CHECK_ARG(comm_out && ipc_handle_out64, "diga_xgmi_create: NULL pointer");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 bytes");
  XgComm* c = new XgComm();
  c->dev.world = world, c->dev.rank = rank, c->dev.max_elems = max_elems;
  void* buf = nullptr;
  // set-up time: the exchange itself never allocates.  Uncached (fine-grained) device memory, as RCCL uses for its own peer
  // buffers: remote writes and local reads do not depend on L2 write-back / invalidation; plain device memory as a fallback.
  hipError_t e = hipExtMallocWithFlags(&buf, xg_total_bytes(c->dev), hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    buf = nullptr;
    e = hipMalloc(&buf, xg_total_bytes(c->dev));
  }
  if (e == hipSuccess) e = hipMemset(buf, 0, xg_total_bytes(c->dev));
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, buf);
  if (e != hipSuccess) {
    diga_set_error("diga_xgmi_create: %s", hipGetErrorString(e));
    if (buf) (void)hipFree(buf);
    delete c;
    return DIGA_ELAUNCH;
  }
  memcpy(ipc_handle_out64, &h, 64);
  c->dev.peer[rank] = (unsigned char*)buf;
  *comm_out = c;
  return DIGA_OK;
}

extern "C" int diga_xgmi_connect(void* comm, const void* handles_world_x64) {
  XgComm* c = (XgComm*)comm;
  DIGA_CHECK_ARG(c && handles_world_x64 && !c->connected, "diga_xgmi_connect: bad communicator");
  for (int r = 0; r < c->dev.world; ++r) {
    if (r == c->dev.rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)handles_world_x64 + 64 * r, 64);
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      diga_set_error("diga_xgmi_connect: hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e));
      return DIGA_ELAUNCH;
    }
    c->opened[r] = p;
    c->dev.peer[r] = (unsigned char*)p;
  }
  c->connected = true;
  return DIGA_OK;
}

extern "C" int diga_xgmi_destroy(void* comm) {
  XgComm* c = (XgComm*)comm;
  if (!c) return DIGA_OK;
  for (int r = 0; r < c->dev.world; ++r)
    if (c->opened[r]) (void)hipIpcCloseMemHandle(c->opened[r]);
  if (c->dev.peer[c->dev.rank]) (void)hipFree(c->dev.peer[c->dev.rank]);
  delete c;
  return DIGA_OK;
}

extern "C" int diga_xgmi_allreduce(void* comm, void* data, int64_t count, int dtype, void* stream) {
  XgComm* c = (XgComm*)comm;
  DIGA_CHECK_ARG(c && c->connected && data, "diga_xgmi_allreduce: communicator not connected / NULL data");
  DIGA_CHECK_ARG(dtype >= 0 && dtype <= 2 && count > 0, "diga_xgmi_allreduce: dtype %d count %lld", dtype, (long long)count);
  DIGA_CHECK_ARG(dtype == 0 || count % 2 == 0, "diga_xgmi_allreduce: 4-byte payloads travel in pairs: count must be even");
  const int64_t n = dtype == 0 ? count : count / 2;       // 8-byte elements
  DIGA_CHECK_ARG(n <= c->dev.max_elems, "diga_xgmi_allreduce: %lld elements exceed the communicator's slot size %lld", (long long)n,
                 (long long)c->dev.max_elems);
  DIGA_CHECK_ARG(((uintptr_t)data & 7) == 0, "diga_xgmi_allreduce: data must be 8-byte aligned");
  const dim3 grid(cdiv(n, XG_EPB)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0)
    hipLaunchKernelGGL(xgmi_allreduce_kernel<0>, grid, block, 0, st, c->dev, (unsigned long long*)data, (long long)n);
  else if (dtype == 1)
    hipLaunchKernelGGL(xgmi_allreduce_kernel<1>, grid, block, 0, st, c->dev, (unsigned long long*)data, (long long)n);
  else
    hipLaunchKernelGGL(xgmi_allreduce_kernel<2>, grid, block, 0, st, c->dev, (unsigned long long*)data, (long long)n);
  DIGA_CHECK_LAUNCH("diga_xgmi_allreduce");
  return DIGA_OK;
}

extern "C" int diga_xgmi_check(void* comm) {
  XgComm* c = (XgComm*)comm;
  DIGA_CHECK_ARG(c != nullptr, "diga_xgmi_check: NULL communicator");
  unsigned err = 0;
  hipError_t e = hipMemcpy(&err, c->dev.peer[c->dev.rank] + xg_error_off(c->dev), 4, hipMemcpyDeviceToHost);   // synchronises the device
  if (e != hipSuccess) {
    diga_set_error("diga_xgmi_check: %s", hipGetErrorString(e));
    return DIGA_ELAUNCH;
  }
  if (err) {
    diga_set_error("diga_xgmi_allreduce: a peer's contribution never arrived (bounded wait expired)");
    return DIGA_ELAUNCH;
  }
  return DIGA_OK;
}
