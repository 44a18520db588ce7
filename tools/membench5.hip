// Follow-up of membench4 on MI355X: (F) what grid shape makes a zero-fill reach the hipMemset rate, (R) how large
// a span must turn from reading to writing at once for an in-place read-modify-write to stop paying for the mix,
// (M) the gradient pass's row mix (runs of RMW rows and runs of zero rows) built from such phases.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));

// F: one-shot fill: wave w writes UNR consecutive KiB
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void k_fill1(f4v* a, size_t n4) {
  const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  f4v* p = a + w * 64 * UNR + (threadIdx.x & 63);
  if ((w + 1) * 64 * UNR > n4) return;
  const f4v z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < UNR; ++k) { if (NT) __builtin_nontemporal_store(z, p + 64 * k); else p[64 * k] = z; }
}
// F: one-shot fill, block-contiguous: thread t of a block writes float4 t, t+256, ... of the block's span
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void k_fill2(f4v* a, size_t n4) {
  f4v* p = a + (size_t)blockIdx.x * 256 * UNR + threadIdx.x;
  if (((size_t)blockIdx.x + 1) * 256 * UNR > n4) return;
  const f4v z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < UNR; ++k) { if (NT) __builtin_nontemporal_store(z, p + 256 * k); else p[256 * k] = z; }
}

// R: block of WAVES waves; per iteration the block reads a contiguous span of WAVES*NL KiB (wave w its NL KiB),
// barrier, writes it back.  Persistent grid.
template <int NL, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_phase(f4v* a, size_t n4) {
  const size_t span = (size_t)64 * NL * WAVES;
  const size_t nspan = n4 / span;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (size_t i = blockIdx.x; i < nspan; i += gridDim.x) {
    f4v* p = a + i * span + (size_t)wv * 64 * NL + lane;
    f4v v[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) v[k] = __builtin_nontemporal_load(p + 64 * k);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NL; ++k) __builtin_nontemporal_store(v[k] * 0.5f, p + 64 * k);
  }
}

// M: rows of 4 KiB, row r of a frame of `period` rows is RMW when r < active, else zero-filled.  A block of WAVES
// waves owns one frame and walks it in chunks of WAVES*RPW rows: loads of the chunk's RMW rows, barrier, all stores.
template <int RPW, int WAVES, bool ZERO_FIRST>
__global__ __launch_bounds__(WAVES * 64) void k_mix(f4v* a, int period, int active) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f4v* blk = a + (size_t)blockIdx.x * period * 256;
  const f4v z = {0.f, 0.f, 0.f, 0.f};
  for (int r0 = 0; r0 < period; r0 += WAVES * RPW) {
    f4v v[RPW][4];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int r = r0 + wv * RPW + j;
      if (r < active) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[j][k] = __builtin_nontemporal_load(blk + (size_t)r * 256 + lane + 64 * k);
      } else if (ZERO_FIRST && r < period) {
#pragma unroll
        for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(z, blk + (size_t)r * 256 + lane + 64 * k);
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int r = r0 + wv * RPW + j;
      if (r < active) {
#pragma unroll
        for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(v[j][k] * 0.5f, blk + (size_t)r * 256 + lane + 64 * k);
      } else if (!ZERO_FIRST && r < period) {
#pragma unroll
        for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(z, blk + (size_t)r * 256 + lane + 64 * k);
      }
    }
  }
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
  const double gb = nfloat * 4 / 1e9;
  printf("== F: one-shot zero-fill of %.2f GB (best of 6)\n", gb);
  { float ms = timeit([&] { CK(hipMemsetAsync(a, 0, nfloat * 4)); }); printf("%-44s %8.3f ms %7.1f GB/s\n", "hipMemsetAsync", ms, gb / ms * 1e3); }
#define F1(U, NT) { unsigned g = (unsigned)(n4 / (256 * U)); float ms = timeit([&] { k_fill1<U, NT><<<g, 256>>>((f4v*)a, n4); }); \
    printf("wave-contiguous %2d KiB/wave %-5s blocks %8u %8.3f ms %7.1f GB/s\n", U, NT ? "nt" : "plain", g, ms, gb / ms * 1e3); }
#define F2(U, NT) { unsigned g = (unsigned)(n4 / (256 * U)); float ms = timeit([&] { k_fill2<U, NT><<<g, 256>>>((f4v*)a, n4); }); \
    printf("block-contiguous %2d KiB/wave %-5s blocks %8u %7.3f ms %7.1f GB/s\n", U, NT ? "nt" : "plain", g, ms, gb / ms * 1e3); }
  F1(1, true) F1(1, false) F1(4, true) F1(4, false) F1(16, true) F1(16, false) F1(64, true) F1(64, false)
  F2(1, true) F2(1, false) F2(4, true) F2(4, false) F2(16, true) F2(16, false) F2(64, true) F2(64, false)
  printf("== R: in-place RMW of %.2f GB; a block reads a span, barrier, writes it (GB/s = read + write)\n", gb);
#define R(NL, W, PERCU) { int g = 256 * PERCU; float ms = timeit([&] { k_phase<NL, W><<<g, W * 64>>>((f4v*)a, n4); }); \
    printf("span %4d KiB = %2d waves x %2d KiB, %d blocks/CU   %8.3f ms %7.1f GB/s\n", NL * W, W, NL, PERCU, ms, 2 * gb / ms * 1e3); }
  R(32, 4, 2) R(32, 4, 3) R(32, 4, 4) R(32, 8, 1) R(32, 8, 2) R(32, 16, 1) R(16, 8, 2) R(16, 8, 4) R(16, 16, 1) R(16, 16, 2)
  R(8, 16, 2) R(8, 16, 4) R(8, 8, 4) R(64, 4, 1) R(64, 4, 2) R(64, 8, 1) R(48, 4, 2) R(48, 8, 1)
  printf("== M: frames of 101 rows x 4 KiB, rows < active RMW, others zero-filled; chunk = waves x rows-per-wave rows per barrier\n");
  const int period = 101; const int nblk = (int)(n4 / (256 * period));
#define M(RPW, W, ZF, ACT) { float ms = timeit([&] { k_mix<RPW, W, ZF><<<nblk, W * 64>>>((f4v*)a, period, ACT); }); \
    printf("active %3d: %2d waves x %d rows, zero rows %-14s %8.3f ms %7.1f GB/s\n", ACT, W, RPW, ZF ? "before barrier" : "after barrier", ms, \
           (double)nblk * (period + ACT) * 4096.0 / ms / 1e6); }
  for (int act : {24, 44, 101}) {
    if (act == 24) { M(8, 4, false, 24) M(8, 4, true, 24) M(4, 8, false, 24) M(4, 16, false, 24) M(7, 16, false, 24) M(2, 16, false, 24) M(8, 8, false, 24) }
    if (act == 44) { M(8, 4, false, 44) M(8, 4, true, 44) M(4, 8, false, 44) M(4, 16, false, 44) M(7, 16, false, 44) M(2, 16, false, 44) M(8, 8, false, 44) }
    if (act == 101) { M(8, 4, false, 101) M(4, 8, false, 101) M(7, 16, false, 101) M(8, 8, false, 101) }
  }
  return 0;
}
