// One-shot all-reduce over direct peer writes: the sharded path's exchanges are tiny and
// latency-bound -- per BN layer [2 groups][2][C] fp64 sums (1..64 KB), per step the prototype sums (311 KB) -- so a ring is
// the wrong algorithm.  Every rank WRITES its contribution straight into a slot of every peer's buffer (one hop over the
// direct xGMI link, peers mapped with hipIpc), raises a flag there, waits for the peers' flags on its own buffer and sums the
// slots in fixed rank order: one kernel launch per exchange, deterministic, identical bits on every rank, capturable in a
// HIP graph (the sequence number lives in device memory).  No counterpart in the reference (it has no distributed code).
//
// Memory model: the buffer is UNCACHED (fine-grained) device memory shared through hipIpc; every access to it -- payload, flags,
// tickets, epochs -- is a system-scope atomic load / store / add (sc0 sc1 on the instruction: it goes past both L2s by itself).
// Ordering between them ("payload before flag", "flag before payload") is then a matter of THIS wave's own operations having
// completed, i.e. `s_waitcnt vmcnt(0)` between the two (xg_order): the stores have been acknowledged by the memory system before
// the flag store issues; a load's data has returned before the loads that depend on its value issue.  The C++ fences that say the
// same thing (release / acquire at system scope) additionally write back and invalidate the WHOLE L2 of the XCD
// (buffer_wbl2 / buffer_inv sc0 sc1) -- for lines that are never cached -- and the L2 is where the convolutions of the other window
// keep their operands: with fences in all 212 exchange launches of a step the sharded step ran 6 % slower than the unsharded one
// (766 vs 814 frames/s, one rank), so they are used only when the buffer could NOT be allocated uncached (`XgDev.fences`; or
// DIGA_XGMI_FENCES=1 in the environment, for A/B runs).  Waiting is a BOUNDED spin: a peer that never arrives makes the
// call fail (error word, DIGA_ELAUNCH from diga_xgmi_check) instead of hanging the GPU.
#include "common.h"
#include "dam_affine.h"
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int XG_MAX_WORLD = 8;
constexpr int XG_MAX_BLOCKS = 512;            // blocks of one launch (each owns XG_EPB elements and its own flag / epoch)
constexpr int XG_EPB = 2048;                  // 8-byte elements per block
constexpr long long XG_TICKS_PER_MS = 100000;  // wall_clock64() counts a constant 100 MHz clock (s_memrealtime)
constexpr int XG_DEFAULT_TIMEOUT_MS = 30000;   // bounded wait: a peer that has not arrived after this long is an error, not a hang
// region B: the exchange fused into the DAM finalize (bn_exchange_finalize_kernel): its own slots, flags, epochs and tickets, so
// that its (channel block, set) element mapping never meets the plain all-reduce's (block, 2048 elements) mapping on one parity
constexpr int XB_MAX_X = 32;                   // 64-channel blocks (C <= 2048)
constexpr int XB_MAX_SETS = 16;                // statistics sets (global batches in flight) per launch
constexpr long long XB_SLOT_ELEMS = (long long)XB_MAX_X * XB_MAX_SETS * 4 * 64;

struct XgDev {                                // passed to the kernel by value
  unsigned char* peer[XG_MAX_WORLD];          // every rank's buffer as mapped in THIS process (peer[rank] = the local one)
  int world, rank;
  long long max_elems;                        // 8-byte elements per slot
  long long timeout_ticks;                    // bound of every wait, in wall_clock64() ticks
  int fences;                                 // 1: system-scope C++ fences around flags (cached buffer / DIGA_XGMI_FENCES=1), 0: xg_order
};

// "everything this wave has issued so far is done" -- the ordering point between payload and flag accesses (see the header)
__device__ __forceinline__ void xg_order(int fences) {
  if (fences)
    __threadfence_system();
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// buffer layout (bytes): [2 parities][world][max_elems] x 8 | flags [world][XG_MAX_BLOCKS] u32 | epoch [XG_MAX_BLOCKS] u32 |
//   region B: [2 parities][world][XB_SLOT_ELEMS] x 8 | flagsB [world][XB_MAX_X] u32 | epochB [XB_MAX_X] u32 | ticketB [XB_MAX_X] u32 | error u32
__host__ __device__ inline size_t xg_slot_off(const XgDev& d, int parity, int r) { return ((size_t)parity * d.world + r) * d.max_elems * 8; }
__host__ __device__ inline size_t xg_flags_off(const XgDev& d) { return (size_t)2 * d.world * d.max_elems * 8; }
__host__ __device__ inline size_t xg_epoch_off(const XgDev& d) { return xg_flags_off(d) + (size_t)d.world * XG_MAX_BLOCKS * 4; }
__host__ __device__ inline size_t xb_slot_off(const XgDev& d, int parity, int r) {
  return xg_epoch_off(d) + (size_t)XG_MAX_BLOCKS * 4 + ((size_t)parity * d.world + r) * XB_SLOT_ELEMS * 8;
}
__host__ __device__ inline size_t xb_flags_off(const XgDev& d) { return xb_slot_off(d, 2, 0); }
__host__ __device__ inline size_t xb_epoch_off(const XgDev& d) { return xb_flags_off(d) + (size_t)d.world * XB_MAX_X * 4; }
__host__ __device__ inline size_t xb_ticket_off(const XgDev& d) { return xb_epoch_off(d) + (size_t)XB_MAX_X * 4; }
__host__ __device__ inline size_t xg_error_off(const XgDev& d) { return xb_ticket_off(d) + (size_t)XB_MAX_X * 4; }

// bounded wait until *flag has reached sequence number e (system-scope acquire loads; ~1 us per poll); false = timed out
// `err` = this rank's sticky error word: once a wait has expired the communicator is dead, and every later wait gives up at once
// (a dead peer costs ONE time-out, not one per exchange of every following step)
__device__ __forceinline__ bool xg_wait_flag(const unsigned* flag, unsigned e, long long timeout_ticks, const unsigned* err, int fences) {
  const long long t0 = wall_clock64();
  while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return false;
    __builtin_amdgcn_s_sleep(8);
    if (wall_clock64() - t0 > timeout_ticks) return false;
  }
  xg_order(fences);          // (the value has been compared, so the load is done; with fences: the acquire)
  return true;
}
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
  xg_order(d.fences);        // every thread: its payload stores have been acknowledged ...
  __syncthreads();           // ... all of the block's, before a flag goes out
  if (tid < d.world) {
    unsigned* f = reinterpret_cast<unsigned*>(d.peer[tid] + xg_flags_off(d)) + (size_t)d.rank * XG_MAX_BLOCKS + b;
    __hip_atomic_store(f, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // 2. wait until every rank's contribution to this block has landed in MY buffer (bounded)
    const unsigned* mine = reinterpret_cast<const unsigned*>(me + xg_flags_off(d)) + (size_t)tid * XG_MAX_BLOCKS + b;
    if (!xg_wait_flag(mine, e, d.timeout_ticks, reinterpret_cast<const unsigned*>(me + xg_error_off(d)), d.fences)) s_fail = 1;
  }
  __syncthreads();
  if (s_fail) {
    // a peer never arrived: do not hang; set the sticky error word (diga_xgmi_check) and POISON the payload -- NaN for the
    // floating-point types, INT32_MIN for counts -- so that the failure travels with the data instead of leaving plausible
    // un-reduced local sums behind
    if (tid == 0) __hip_atomic_store(reinterpret_cast<unsigned*>(me + xg_error_off(d)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long poison = DT == 0 ? 0x7ff8000000000000ull : DT == 1 ? 0x7fc000007fc00000ull : 0x8000000080000000ull;
    for (long long i = lo + tid; i < hi; i += 256) data[i] = poison;
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
  xg_order(d.fences);        // the slots have been read: a peer that sees the next flag may overwrite this parity two exchanges on
  __syncthreads();
  if (tid == 0) epoch[b] = e;
}

// The exchange FUSED into the DAM finalize: ONE launch per BN layer does what diga_bn_reduce_slab +
// diga_xgmi_allreduce + diga_bn_finalize_sums do in three.  Grid (C / 64, nsets), block = 64 channels x FIN_PH unit phases.
//   1. every block reduces its (channel block x, set) of the local slab in the fixed order of diga_bn_finalize (dam_affine.h);
//   2. its wave 0 writes the 4 x 64 fp64 group sums into slot [parity][rank] of EVERY rank's buffer (one hop over the direct
//      link), fences, and takes a ticket of channel block x;
//   3. the LAST of the nsets blocks of x (all of this rank's sums of x have landed everywhere) raises flag [rank][x] on every
//      peer, waits -- bounded -- for the peers' flags on its own buffer, sums the slots in RANK ORDER (identical bits on every
//      rank, and the bits of diga_xgmi_allreduce) and turns them into scale / shift [nsets][C] with dam_affine_from_sums.
// Only C / 64 <= 32 blocks of a launch ever wait for a peer, so two communicators in flight on two HIP streams (two windows)
// cannot starve each other's convolutions of CUs: every other block has finished once its sums are written.
// Time-out: scale / shift of the channel block are set to NaN and the sticky error word is raised.
__global__ __launch_bounds__(64 * FIN_PH) void bn_exchange_finalize_kernel(const XgDev d, const float* __restrict__ slab, int ups,
                                                                          int nsamples, int a_samples, int per_sample, int C, double n_a,
                                                                          double n_b, const float* __restrict__ rmean,
                                                                          const float* __restrict__ rvar, const float* __restrict__ weight,
                                                                          const float* __restrict__ bias, float lam, float eps,
                                                                          float* __restrict__ scale, float* __restrict__ shift) {
  __shared__ double red[FIN_PH][4][64];
  __shared__ int s_last, s_fail;
  const int tid = threadIdx.x, lane = tid & 63, ph = tid >> 6;
  const int x = blockIdx.x, set = blockIdx.y, nsets = gridDim.y;
  const int c = x * 64 + lane;
  unsigned char* const me = d.peer[d.rank];
  unsigned* const epoch = reinterpret_cast<unsigned*>(me + xb_epoch_off(d)) + x;
  unsigned* const ticket = reinterpret_cast<unsigned*>(me + xb_ticket_off(d)) + x;
  const unsigned e = __hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1u;     // the same on every rank
  const int parity = (int)(e & 1u);
  if (tid == 0) s_last = 0, s_fail = 0;
  double s[4];
  if (per_sample) {     // set = local sample = this rank's frame of global batch `set`: all of it is group A (rank 0) or group B
    const int u0 = set * ups, u1 = u0 + ups;
    slab_group_sums(slab, u0, a_samples > 0 ? u1 : u0, u1, C, c, c < C, red, s);
  } else {
    slab_group_sums(slab, 0, a_samples * ups, nsamples * ups, C, c, c < C, red, s);
  }
  const size_t elem0 = ((size_t)(x * XB_MAX_SETS + set) * 4) * 64 + lane;
  if (ph == 0) {
    for (int r = 0; r < d.world; ++r) {
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(d.peer[r] + xb_slot_off(d, parity, d.rank)) + elem0;
#pragma unroll
      for (int k = 0; k < 4; ++k) __hip_atomic_store(dst + k * 64, Acc<double>::to_bits(s[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    xg_order(d.fences);      // wave 0 wrote them all: acknowledged before the ticket is taken
  }
  __syncthreads();
  if (tid == 0) {
    s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == (unsigned)(nsets - 1);
    xg_order(d.fences);      // the ticket's value is here: every earlier taker's sums had been acknowledged before its add
  }
  __syncthreads();
  if (!s_last) return;
  if (tid < d.world) {
    unsigned* f = reinterpret_cast<unsigned*>(d.peer[tid] + xb_flags_off(d)) + (size_t)d.rank * XB_MAX_X + x;
    __hip_atomic_store(f, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned* mine = reinterpret_cast<const unsigned*>(me + xb_flags_off(d)) + (size_t)tid * XB_MAX_X + x;
    if (!xg_wait_flag(mine, e, d.timeout_ticks, reinterpret_cast<const unsigned*>(me + xg_error_off(d)), d.fences)) s_fail = 1;
  }
  __syncthreads();
  if (ph < nsets && c < C) {          // thread = (set ph, channel lane)
    float* sc = scale + (size_t)ph * C;
    float* sh = shift + (size_t)ph * C;
    if (s_fail) {
      sc[c] = sh[c] = __uint_as_float(0x7fc00000u);
    } else {
      const size_t el = ((size_t)(x * XB_MAX_SETS + ph) * 4) * 64 + lane;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double v = 0.0;
        for (int r = 0; r < d.world; ++r)
          v += Acc<double>::from_bits(__hip_atomic_load(reinterpret_cast<const unsigned long long*>(me + xb_slot_off(d, parity, r)) + el + k * 64,
                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        s[k] = v;
      }
      dam_affine_from_sums(s, c, n_a, n_b, rmean, rvar, weight, bias, lam, eps, sc, sh);
    }
  }
  xg_order(d.fences);        // the slots have been read before the epoch moves on
  __syncthreads();
  if (tid == 0) {
    if (s_fail) __hip_atomic_store(reinterpret_cast<unsigned*>(me + xg_error_off(d)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(epoch, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
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
  c->dev.timeout_ticks = (long long)XG_DEFAULT_TIMEOUT_MS * XG_TICKS_PER_MS;
  void* buf = nullptr;
  // set-up time: the exchange itself never allocates.  Uncached (fine-grained) device memory, as RCCL uses for its own peer
  // buffers: remote writes and local reads do not depend on L2 write-back / invalidation; plain device memory as a fallback.
  hipError_t e = hipExtMallocWithFlags(&buf, xg_total_bytes(c->dev), hipDeviceMallocUncached);
  const char* env = getenv("DIGA_XGMI_FENCES");
  c->dev.fences = env && env[0] == '1';
  if (e != hipSuccess) {
    (void)hipGetLastError();
    buf = nullptr;
    e = hipMalloc(&buf, xg_total_bytes(c->dev));
    c->dev.fences = 1;        // a cached buffer: lines of it may sit in an L2, only the fences' write-back / invalidate reach them
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
    diga_set_error("one-shot exchange: a peer's contribution never arrived (bounded wait expired); the payload was poisoned (NaN / INT32_MIN) "
                   "and the communicator must not be used again");
    return DIGA_ELAUNCH;
  }
  return DIGA_OK;
}

extern "C" int diga_xgmi_set_timeout_ms(void* comm, int timeout_ms) {
  XgComm* c = (XgComm*)comm;
  DIGA_CHECK_ARG(c != nullptr && timeout_ms >= 1 && timeout_ms <= 600000, "diga_xgmi_set_timeout_ms: NULL communicator or %d ms outside [1, 600000]",
                 timeout_ms);
  c->dev.timeout_ticks = (long long)timeout_ms * XG_TICKS_PER_MS;
  return DIGA_OK;
}

extern "C" int diga_bn_exchange_finalize(void* comm, const float* slab, int units_per_sample, int nsamples, int a_samples, int per_sample,
                                         int C, double n_a, double n_b, const float* running_mean, const float* running_var,
                                         const float* weight, const float* bias, float lam, float eps, float* scale, float* shift,
                                         void* stream) {
  XgComm* c = (XgComm*)comm;
  DIGA_CHECK_ARG(c && c->connected, "diga_bn_exchange_finalize: communicator not connected");
  DIGA_CHECK_ARG(slab && running_mean && running_var && weight && bias && scale && shift, "diga_bn_exchange_finalize: NULL pointer");
  DIGA_CHECK_ARG(units_per_sample > 0 && nsamples > 0 && a_samples >= 0 && a_samples <= nsamples && n_a > 0 && n_b >= 0,
                 "diga_bn_exchange_finalize: bad sizes units_per_sample=%d nsamples=%d a_samples=%d", units_per_sample, nsamples, a_samples);
  DIGA_CHECK_ARG(C >= 64 && C % 64 == 0 && C / 64 <= XB_MAX_X, "diga_bn_exchange_finalize: C=%d must be a multiple of 64 up to %d", C, 64 * XB_MAX_X);
  const int nsets = per_sample ? nsamples : 1;
  DIGA_CHECK_ARG(nsets <= XB_MAX_SETS && nsets <= FIN_PH, "diga_bn_exchange_finalize: %d sets in flight exceed the limit %d", nsets,
                 XB_MAX_SETS < FIN_PH ? XB_MAX_SETS : FIN_PH);
  hipLaunchKernelGGL(bn_exchange_finalize_kernel, dim3(C / 64, nsets), dim3(64 * FIN_PH), 0, (hipStream_t)stream, c->dev, slab, units_per_sample,
                     nsamples, a_samples, per_sample, C, n_a, n_b, running_mean, running_var, weight, bias, lam, eps, scale, shift);
  DIGA_CHECK_LAUNCH("diga_bn_exchange_finalize");
  return DIGA_OK;
}
