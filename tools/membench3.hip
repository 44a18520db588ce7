// Does the 256 MiB memory-side cache serve the second read of a chunk?  Pass A reads a chunk, pass B
// rewrites it in place; chunks are walked over a 10 GiB buffer.  Reports total time per sweep of the buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void k_read(const f4v* a, size_t n4, float* sink) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t ntiles = n4 / 256;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 4 + wv; i < ntiles; i += (size_t)gridDim.x * 4) {
    const f4v* p = a + i * 256 + lane;
    f4v v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = NT ? __builtin_nontemporal_load(p + 64 * k) : p[64 * k];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
  }
  if (acc == 1234.5f) sink[0] = acc;
}
template <bool NTL, bool NTS, bool REV>
__global__ __launch_bounds__(256) void k_rmw(f4v* a, size_t n4) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t ntiles = n4 / 256;
  for (size_t i0 = (size_t)blockIdx.x * 4 + wv; i0 < ntiles; i0 += (size_t)gridDim.x * 4) {
    const size_t i = REV ? ntiles - 1 - i0 : i0;
    f4v* p = a + i * 256 + lane;
    f4v v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = NTL ? __builtin_nontemporal_load(p + 64 * k) : p[64 * k];
#pragma unroll
    for (int k = 0; k < 4; ++k) { f4v o = v[k] * 0.5f; if (NTS) __builtin_nontemporal_store(o, p + 64 * k); else p[64 * k] = o; }
  }
}

template <bool NTA, bool NTL, bool NTS, bool REV>
void run(const char* name, f4v* a, size_t n4, size_t chunk_mb, float* sink, int blocks) {
  const size_t c4 = chunk_mb * (1ull << 20) / 16;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    for (size_t o = 0; o + c4 <= n4; o += c4) {
      k_read<NTA><<<blocks, 256>>>(a + o, c4, sink);
      k_rmw<NTL, NTS, REV><<<blocks, 256>>>(a + o, c4);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double bytes = (double)(n4 / c4 * c4) * 16;
  printf("%-40s chunk=%5zu MiB  %8.3f ms  algorithmic(2x) %7.1f GB/s  traffic-model(3x) %7.1f GB/s\n", name, chunk_mb, best,
         2 * bytes / best / 1e6, 3 * bytes / best / 1e6);
}

int main() {
  const size_t n4 = 10ull * (1ull << 30) / 16;
  f4v* a; float* sink; CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&sink, 64)); CK(hipMemset(a, 0x3c, n4 * 16));
  for (size_t mb : {32, 64, 96, 128, 160, 192, 224, 256, 320, 512, 10240}) {
    run<false, false, false, false>("A plain | B plain fwd", a, n4, mb, sink, 2048);
    run<false, false, false, true>("A plain | B plain rev", a, n4, mb, sink, 2048);
    run<false, true, true, true>("A plain | B nt rev", a, n4, mb, sink, 2048);
    run<true, true, true, true>("A nt    | B nt rev", a, n4, mb, sink, 2048);
  }
  return 0;
}
