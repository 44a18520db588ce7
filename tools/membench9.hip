// In-place read-modify-write of a large buffer by a resident grid whose blocks read a span, meet at a barrier and
// write it back (the gradient pass's k_grad_rmw): does it pay to make ALL blocks of the device turn from reading to
// writing together, so that the memory sees long runs of reads and long runs of writes instead of a mixture?
//   mode 0  blocks run free (block barrier only)
//   mode 1  device-wide barrier between a round's reads and its writes (every block resident; bounded spin)
//   mode 2  device-wide barrier at both turns
//   mode 3  phases by the wall clock: reads may only be issued in even slices of `slice` ticks, writes in odd ones
// usage: membench9 [GB, default 10.6]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ unsigned int g_ctr;
__device__ unsigned int g_bail;

__device__ __forceinline__ void device_barrier(unsigned int target) {
  // thread 0 of every block; the counter only grows (one launch = a known number of arrivals)
  __hip_atomic_fetch_add(&g_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(&g_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(2);
    if (wall_clock64() - t0 > 100000ull) {  // 1 ms of the 100 MHz clock: give up rather than hang
      atomicAdd(&g_bail, 1u);
      break;
    }
  }
}

template <int NL, int WAVES, int MODE>
__global__ __launch_bounds__(WAVES * 64) void k_phase(f4v* a, size_t n4, unsigned int base, int slice) {
  const size_t span = (size_t)64 * NL * WAVES;
  const size_t nspan = n4 / span;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t rounds = (nspan + gridDim.x - 1) / gridDim.x;
  for (size_t r = 0; r < rounds; ++r) {
    const size_t i = r * gridDim.x + blockIdx.x;
    const bool live = i < nspan;
    f4v* p = a + (live ? i : 0) * span + (size_t)wv * 64 * NL + lane;
    f4v v[NL];
    if (MODE == 3) {
      if (threadIdx.x == 0) while (((wall_clock64() / slice) & 1ull) != 0ull) __builtin_amdgcn_s_sleep(1);
      __syncthreads();
    }
    if (MODE == 2) {
      if (threadIdx.x == 0) device_barrier(base + (unsigned)(2 * r + 1) * gridDim.x);
      __syncthreads();
    }
    if (live) {
#pragma unroll
      for (int k = 0; k < NL; ++k) v[k] = __builtin_nontemporal_load(p + 64 * k);
    }
    if (MODE != 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < NL; ++k) asm volatile("" : "+v"(v[k]));
    }
    __syncthreads();
    if (MODE == 1 || MODE == 2) {
      if (threadIdx.x == 0) device_barrier(base + (unsigned)((MODE == 2 ? 2 * r + 2 : r + 1)) * gridDim.x);
      __syncthreads();
    }
    if (MODE == 3) {
      if (threadIdx.x == 0) while (((wall_clock64() / slice) & 1ull) != 1ull) __builtin_amdgcn_s_sleep(1);
      __syncthreads();
    }
    if (live) {
#pragma unroll
      for (int k = 0; k < NL; ++k) __builtin_nontemporal_store(v[k] * 0.5f, p + 64 * k);
    }
  }
}

int main(int argc, char** argv) {
  const double want_gb = argc > 1 ? atof(argv[1]) : 10.6;
  const size_t n4 = (size_t)(want_gb * 1e9 / 16);
  float* a; CK(hipMalloc(&a, n4 * 16)); CK(hipMemset(a, 0x3c, n4 * 16));
  const double gb = n4 * 16 / 1e9;
  unsigned int zero = 0;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_ctr), &zero, 4)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_bail), &zero, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned int base = 0;
#define RUN(NL, W, MODE, PERCU, SLICE) { \
    const int g = 256 * PERCU; \
    const size_t nspan = n4 / ((size_t)64 * NL * W); const unsigned rounds = (unsigned)((nspan + g - 1) / g); \
    float best = 1e30f; \
    for (int rep = 0; rep < 4; ++rep) { \
      CK(hipEventRecord(e0)); k_phase<NL, W, MODE><<<g, W * 64>>>((f4v*)a, n4, base, SLICE); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
      if (MODE == 1) base += rounds * g; if (MODE == 2) base += 2 * rounds * g; \
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } \
    unsigned int bail; CK(hipMemcpyFromSymbol(&bail, HIP_SYMBOL(g_bail), 4)); \
    printf("mode %d  %2d waves x %2d KiB, %d blocks/CU, slice %5d   %8.3f ms %7.1f GB/s   (bails %u)\n", MODE, W, NL, PERCU, SLICE, best, 2 * gb / best * 1e3, bail); }
  printf("in-place RMW of %.2f GB (GB/s = read + write)\n", gb);
  RUN(32, 6, 0, 2, 0)
  RUN(32, 4, 0, 2, 0) RUN(32, 4, 1, 2, 0) RUN(32, 4, 1, 3, 0) RUN(32, 4, 2, 3, 0)
  RUN(32, 8, 0, 1, 0) RUN(32, 8, 1, 1, 0)
  RUN(16, 8, 1, 2, 0) RUN(16, 8, 1, 4, 0)
  RUN(32, 4, 3, 2, 300) RUN(32, 4, 3, 2, 600) RUN(32, 4, 3, 2, 900)
  return 0;
}
