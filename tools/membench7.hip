// MI355X: (A) does throttling a wave's outstanding stores make a zero-fill faster (a fill in which every wave writes
// one KiB and ends runs at 6.9 TB/s, one in which a wave issues 4-32 KiB back to back at 5.3-6.3)?
// (B) read-only streams: one-shot blocks whose waves own contiguous spans, against row-interleaved waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));

template <int K> __device__ __forceinline__ void wait_vm() {
  if constexpr (K == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (K == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (K == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (K == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (K == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

// one-shot: block of WAVES waves, wave w writes KIB consecutive KiB; after every store at most K stay outstanding (K < 0: free)
template <int KIB, int WAVES, int K>
__global__ __launch_bounds__(WAVES * 64) void k_fill(f4v* a, size_t n4) {
  const size_t w = (size_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if ((w + 1) * 64 * KIB > n4) return;
  f4v* p = a + w * 64 * KIB + (threadIdx.x & 63);
  const f4v z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < KIB; ++k) { __builtin_nontemporal_store(z, p + 64 * k); wait_vm<K>(); }
}

// read-only, one-shot, wave-contiguous spans of KIB KiB; sum to keep the loads alive
template <int KIB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_read_span(const f4v* a, size_t n4, float* out) {
  const size_t w = (size_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if ((w + 1) * 64 * KIB > n4) return;
  const f4v* p = a + w * 64 * KIB + (threadIdx.x & 63);
  f4v v[KIB];
#pragma unroll
  for (int k = 0; k < KIB; ++k) v[k] = __builtin_nontemporal_load(p + 64 * k);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < KIB; ++k) s += v[k].x + v[k].y + v[k].z + v[k].w;
  if (s == 123.456f) out[0] = s;
}
// read-only, block per `rows` rows of 4 KiB, waves interleave rows, two rows in flight per wave (the first pass's shape)
__global__ __launch_bounds__(256) void k_read_rows(const f4v* a, int rows, float* out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const f4v* blk = a + (size_t)blockIdx.x * rows * 256;
  float s = 0.f;
  f4v va[4], vb[4];
  int r = wv;
  if (r < rows) for (int k = 0; k < 4; ++k) va[k] = __builtin_nontemporal_load(blk + (size_t)r * 256 + lane + 64 * k);
  for (; r < rows; r += 8) {
    if (r + 4 < rows) for (int k = 0; k < 4; ++k) vb[k] = __builtin_nontemporal_load(blk + (size_t)(r + 4) * 256 + lane + 64 * k);
    for (int k = 0; k < 4; ++k) s += va[k].x + va[k].y + va[k].z + va[k].w;
    if (r + 8 < rows) for (int k = 0; k < 4; ++k) va[k] = __builtin_nontemporal_load(blk + (size_t)(r + 8) * 256 + lane + 64 * k);
    if (r + 4 < rows) for (int k = 0; k < 4; ++k) s += vb[k].x + vb[k].y + vb[k].z + vb[k].w;
  }
  if (s == 123.456f) out[0] = s;
}

template <typename F>
float timeit(F launch, int reps = 6) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); launch();
  float best = 1e30f;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return best;
}

int main() {
  const size_t nfloat = 10ull * (1ull << 30) / 4, n4 = nfloat / 4;
  float* a; CK(hipMalloc(&a, nfloat * 4)); CK(hipMemset(a, 0x3c, nfloat * 4));
  float* out; CK(hipMalloc(&out, 16));
  const double gb = nfloat * 4 / 1e9;
  printf("== A: one-shot zero-fill of %.2f GB, wave-contiguous, outstanding stores per wave limited to K\n", gb);
#define A(KIB, W, K) { unsigned g = (unsigned)(n4 / (64 * KIB * W)); float ms = timeit([&] { k_fill<KIB, W, K><<<g, W * 64>>>((f4v*)a, n4); }); \
    printf("%2d KiB/wave %2d waves/block  K=%2d   blocks %8u %8.3f ms %7.1f GB/s\n", KIB, W, K, g, ms, gb / ms * 1e3); }
  A(1, 4, -1) A(1, 8, -1) A(1, 16, -1) A(2, 4, -1) A(2, 4, 0)
  A(4, 4, -1) A(4, 4, 0) A(4, 4, 1) A(4, 4, 2)
  A(8, 4, -1) A(8, 4, 0) A(8, 4, 1) A(8, 4, 2) A(8, 4, 4)
  A(32, 8, -1) A(32, 8, 0) A(32, 8, 1) A(32, 8, 2) A(32, 8, 4) A(32, 8, 8)
  A(32, 6, -1) A(32, 6, 1) A(32, 6, 4)
  printf("== B: read-only stream of %.2f GB\n", gb);
#define B(KIB, W) { unsigned g = (unsigned)(n4 / (64 * KIB * W)); float ms = timeit([&] { k_read_span<KIB, W><<<g, W * 64>>>((const f4v*)a, n4, out); }); \
    printf("span %2d KiB/wave %2d waves/block     blocks %8u %8.3f ms %7.1f GB/s\n", KIB, W, g, ms, gb / ms * 1e3); }
  B(4, 4) B(8, 4) B(16, 4) B(32, 4) B(8, 8) B(16, 8) B(32, 8) B(16, 16) B(32, 6) B(16, 6)
  { const int rows = 101; unsigned g = (unsigned)(n4 / (256 * rows)); float ms = timeit([&] { k_read_rows<<<g, 256>>>((const f4v*)a, rows, out); });
    printf("rows interleaved over 4 waves, 2 in flight, block = 101 rows  %8.3f ms %7.1f GB/s\n", ms, (double)g * rows * 4096 / ms / 1e6); }
  return 0;
}
