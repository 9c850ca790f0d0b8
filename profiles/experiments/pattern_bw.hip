// Access-pattern bandwidth of the fused-residual-input conv1's operand stream (experiment, not product code).
// A block of 512 threads owns 128 rows of two bf16 [M][C] tensors (C = 1024: 2048-byte rows), reads both and writes one, as
// that kernel does, 16 KB per tensor per step, four steps of loads in flight, one block per CU (LDS-limited like the kernel).
// SEG = contiguous bytes one step takes from a row: 128 is the kernel's pattern (a 64-channel K step of every row);
// 256 ... 2048 take fewer rows per step and more of each.   hipcc -O3 --offload-arch=gfx950 pattern_bw.hip -o pattern_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int SEG, bool WB>
__global__ __launch_bounds__(512) void stream_kernel(const char* __restrict__ a, const char* __restrict__ b, char* __restrict__ o, int M, int pitch) {
  extern __shared__ char lds[];
  constexpr int LPR = SEG / 16;          // lanes per row segment
  constexpr int RPS = 8192 / SEG;        // rows one 512-thread load instruction covers
  const int tid = threadIdx.x;
  const int row0 = blockIdx.x * 128;
  const int segs_per_row = pitch / SEG;
  const int steps = (128 / RPS) * segs_per_row / 2;   // two load instructions per tensor per step (16 KB)
  u32x4 acc = {0, 0, 0, 0};
  // instruction index q -> (segment of the row, row group): row groups vary fastest, so that all 128 rows supply a K range before the next
  auto addr = [&](int q) {
    const int grp = q % (128 / RPS), seg = q / (128 / RPS);
    int row = row0 + grp * RPS + tid / LPR;
    row = row < M ? row : M - 1;
    return (size_t)row * pitch + seg * SEG + (tid % LPR) * 16;
  };
  for (int t = 0; t < steps; t += 4) {
    u32x4 va[8], vb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = min(t * 2 + i, steps * 2 - 1);
      va[i] = *reinterpret_cast<const u32x4*>(a + addr(q));
      vb[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(b + addr(q)));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = min(t * 2 + i, steps * 2 - 1);
      const u32x4 v = va[i] ^ vb[i];
      if (WB) *reinterpret_cast<u32x4*>(o + addr(q)) = v; else acc ^= v;
    }
  }
  if (!WB && acc[0] == 0x12345678u) *reinterpret_cast<u32x4*>(o) = acc;
  if (tid == 100000) lds[0] = 0;
}

template <int SEG, bool WB>
void run(const char* a, const char* b, char* o, int M, int pitch) {
  const int blocks = (M + 127) / 128;
  const int smem = 96 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<SEG, WB>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<SEG, WB>), dim3(blocks), dim3(512), smem, 0, a, b, o, M, pitch);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((stream_kernel<SEG, WB>), dim3(blocks), dim3(512), smem, 0, a, b, o, M, pitch);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)M * pitch * (WB ? 3 : 2);
  printf("SEG %4d  write-back %d: %7.1f us  %6.0f GB/s\n", SEG, (int)WB, ms * 100, bytes / (ms / 10 * 1e-3) / 1e9);
}


// The same stream software-pipelined as the conv kernel's K loop is: the loads of steps t + 1 ... t + DEPTH are in flight while
// step t is combined and written back (one step = 2 loads per tensor and 2 stores per thread).
template <int DEPTH>
__global__ __launch_bounds__(512) void pipe_kernel(const char* __restrict__ a, const char* __restrict__ b, char* __restrict__ o, int M, int pitch) {
  extern __shared__ char lds[];
  const int tid = threadIdx.x;
  const int row0 = blockIdx.x * 128;
  const int steps = pitch / 128;
  auto addr = [&](int t, int i) {
    int row = row0 + i * 64 + tid / 8;
    row = row < M ? row : M - 1;
    return (size_t)row * pitch + t * 128 + (tid % 8) * 16;
  };
  u32x4 va[DEPTH + 1][2], vb[DEPTH + 1][2];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      va[d][i] = *reinterpret_cast<const u32x4*>(a + addr(d, i));
      vb[d][i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(b + addr(d, i)));
    }
  for (int t0 = 0; t0 < steps; t0 += DEPTH + 1) {
#pragma unroll
    for (int u = 0; u <= DEPTH; ++u) {
      const int t = t0 + u;
      if (t < steps) {
        const int tl = t + DEPTH;
        if (tl < steps) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            va[(u + DEPTH) % (DEPTH + 1)][i] = *reinterpret_cast<const u32x4*>(a + addr(tl, i));
            vb[(u + DEPTH) % (DEPTH + 1)][i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(b + addr(tl, i)));
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(o + addr(t, i)) = va[u][i] ^ vb[u][i];
      }
    }
  }
  if (tid == 100000) lds[0] = 0;
}

template <int DEPTH>
void run_pipe(const char* a, const char* b, char* o, int M, int pitch, int smem) {
  const int blocks = (M + 127) / 128;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pipe_kernel<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((pipe_kernel<DEPTH>), dim3(blocks), dim3(512), smem, 0, a, b, o, M, pitch);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((pipe_kernel<DEPTH>), dim3(blocks), dim3(512), smem, 0, a, b, o, M, pitch);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)M * pitch * 3;
  printf("pipelined, %d steps in flight, %3d KB LDS per block: %7.1f us  %6.0f GB/s\n", DEPTH, smem / 1024, ms * 100, bytes / (ms / 10 * 1e-3) / 1e9);
}

// bn_apply's launch shape on the same stream: 256 threads own RPB whole rows (C = 1024: 128 lanes per row), every thread one
// 16-byte chunk of each of its rows; blocks walk the tensor linearly.
template <int RPT>
__global__ __launch_bounds__(256) void rows_kernel(const char* __restrict__ a, const char* __restrict__ b, char* __restrict__ o, int M, int pitch) {
  const int lpr = pitch / 16, rstep = 256 / lpr;
  const int row0 = blockIdx.x * rstep * RPT + threadIdx.x / lpr;
  u32x4 va[RPT], vb[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int row = min(row0 + i * rstep, M - 1);
    const size_t ad = (size_t)row * pitch + (threadIdx.x % lpr) * 16;
    va[i] = *reinterpret_cast<const u32x4*>(a + ad);
    vb[i] = *reinterpret_cast<const u32x4*>(b + ad);
  }
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int row = min(row0 + i * rstep, M - 1);
    const size_t ad = (size_t)row * pitch + (threadIdx.x % lpr) * 16;
    *reinterpret_cast<u32x4*>(o + ad) = va[i] ^ vb[i];
  }
}

template <int RPT>
void run_rows(const char* a, const char* b, char* o, int M, int pitch) {
  const int rpb = 256 / (pitch / 16) * RPT, blocks = (M + rpb - 1) / rpb;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((rows_kernel<RPT>), dim3(blocks), dim3(256), 0, 0, a, b, o, M, pitch);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((rows_kernel<RPT>), dim3(blocks), dim3(256), 0, 0, a, b, o, M, pitch);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("whole rows, %2d rows per thread, %6d blocks of 256: %7.1f us  %6.0f GB/s\n", RPT, blocks, ms * 100, (double)M * pitch * 3 / (ms / 10 * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
  const int M = 15 * 65 * 129, C = argc > 1 ? atoi(argv[1]) : 1024, pitch = C * 2;
  const size_t n = (size_t)M * pitch;
  char *a, *b, *o;
  CK(hipMalloc(&a, n)); CK(hipMalloc(&b, n)); CK(hipMalloc(&o, n));
  CK(hipMemset(a, 1, n)); CK(hipMemset(b, 2, n)); CK(hipMemset(o, 0, n));
  printf("M %d  C %d  (%.0f MB per tensor)\n", M, C, n / 1e6);
  run<128, false>(a, b, o, M, pitch); run<128, true>(a, b, o, M, pitch);
  run<256, false>(a, b, o, M, pitch); run<256, true>(a, b, o, M, pitch);
  run<512, false>(a, b, o, M, pitch); run<512, true>(a, b, o, M, pitch);
  if (pitch >= 1024) { run<1024, false>(a, b, o, M, pitch); run<1024, true>(a, b, o, M, pitch); }
  if (pitch >= 2048) { run<2048, false>(a, b, o, M, pitch); run<2048, true>(a, b, o, M, pitch); }
  run_rows<1>(a, b, o, M, pitch); run_rows<2>(a, b, o, M, pitch); run_rows<4>(a, b, o, M, pitch); run_rows<16>(a, b, o, M, pitch);
  run_pipe<1>(a, b, o, M, pitch, 96 * 1024); run_pipe<2>(a, b, o, M, pitch, 96 * 1024); run_pipe<3>(a, b, o, M, pitch, 96 * 1024);
  run_pipe<2>(a, b, o, M, pitch, 72 * 1024); run_pipe<2>(a, b, o, M, pitch, 48 * 1024); run_pipe<2>(a, b, o, M, pitch, 1024);
  CK(hipDeviceSynchronize());
  return 0;
}
