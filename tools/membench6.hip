// Follow-up of membench5 on MI355X: the gradient pass's row mix (runs of RMW rows and runs of zero rows, 4 KiB
// rows, 101 rows per frame) done by ONE-SHOT blocks of WAVES waves x RPW rows each (block b owns rows
// b*WAVES*RPW ...): loads of the block's RMW rows, optional barrier, all stores.  How small / large should the
// block's span be, and does the barrier matter?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));

template <int RPW, int WAVES, bool SYNC, bool ROWMAJOR>
__global__ __launch_bounds__(WAVES * 64) void k_mix1(f4v* a, long long nrows, int period, int active) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.x * (WAVES * RPW);
  const f4v z = {0.f, 0.f, 0.f, 0.f};
  f4v v[RPW][4];
  bool act[RPW];
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    // ROWMAJOR: wave w owns rows w*RPW.. (RPW consecutive rows = RPW*4 KiB contiguous); else rows w, w+WAVES, ...
    const long long r = r0 + (ROWMAJOR ? wv * RPW + j : j * WAVES + wv);
    act[j] = r < nrows && (int)(r % period) < active;
    if (act[j]) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[j][k] = __builtin_nontemporal_load(a + r * 256 + lane + 64 * k);
    }
  }
  if (SYNC) __syncthreads();
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const long long r = r0 + (ROWMAJOR ? wv * RPW + j : j * WAVES + wv);
    if (r >= nrows) continue;
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(act[j] ? v[j][k] * 0.5f : z, a + r * 256 + lane + 64 * k);
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
  const int period = 101;
  const long long nrows = (long long)(n4 / 256);
#define M(RPW, W, SYNC, RM, ACT) { unsigned g = (unsigned)((nrows + W * RPW - 1) / (W * RPW)); \
    float ms = timeit([&] { k_mix1<RPW, W, SYNC, RM><<<g, W * 64>>>((f4v*)a, nrows, period, ACT); }); \
    printf("active %3d: %2d waves x %d rows (%4d KiB) %-7s %-9s blocks %8u %8.3f ms %7.1f GB/s\n", ACT, W, RPW, W * RPW * 4, \
           SYNC ? "barrier" : "free", RM ? "row-major" : "interleav", g, ms, (double)nrows * (period + ACT) / period * 4096.0 / ms / 1e6); }
#define SET(ACT) \
  M(1, 4, false, false, ACT) M(1, 4, true, false, ACT) M(2, 4, false, false, ACT) M(2, 4, true, false, ACT) M(4, 4, false, false, ACT) M(4, 4, true, false, ACT) \
  M(8, 4, false, false, ACT) M(8, 4, true, false, ACT) M(8, 4, true, true, ACT) M(1, 8, true, false, ACT) M(2, 8, true, false, ACT) M(4, 8, true, false, ACT) M(8, 8, true, false, ACT) \
  M(8, 8, true, true, ACT) M(1, 16, true, false, ACT) M(2, 16, true, false, ACT) M(4, 16, true, false, ACT) M(4, 16, false, false, ACT) M(8, 16, true, false, ACT)
  SET(0) SET(44) SET(101)
  return 0;
}
