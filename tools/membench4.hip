// Write-stream and read/write-phase experiments on MI355X (one 10 GiB buffer):
//  W: write-only fills: bytes per lane per store (4 / 8 / 16), plain or non-temporal, grid shape
//  P: in-place read-modify-write with the reads and the writes of a block separated in time: a block loads a
//     tile of KB KiB into registers (all loads first), waits, then stores it all; against the interleaved form
//  M: the gradient pass's mix: runs of RMW rows alternating with runs of zero-filled rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

// ---- W: each wave fills consecutive 1 KiB-per-instruction pieces (16 B per lane) or narrower ones
template <int WIDTH, bool NT, int UNR>
__global__ __launch_bounds__(256) void k_fill(float* a, size_t nfloat) {
  const size_t per_wave_iter = (size_t)64 * (WIDTH / 4) * UNR;  // floats per wave per iteration
  const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
  const int lane = threadIdx.x & 63;
  for (size_t base = gw * per_wave_iter; base < nfloat; base += nw * per_wave_iter) {
#pragma unroll
    for (int k = 0; k < UNR; ++k) {
      float* p = a + base + (size_t)k * 64 * (WIDTH / 4) + (size_t)lane * (WIDTH / 4);
      if constexpr (WIDTH == 16) { f4v z = {0.f, 0.f, 0.f, 0.f}; if (NT) __builtin_nontemporal_store(z, (f4v*)p); else *(f4v*)p = z; }
      else if constexpr (WIDTH == 8) { f2v z = {0.f, 0.f}; if (NT) __builtin_nontemporal_store(z, (f2v*)p); else *(f2v*)p = z; }
      else { if (NT) __builtin_nontemporal_store(0.f, p); else *p = 0.f; }
    }
  }
}

// ---- P: tile of NL float4 per lane per wave: all loads, wait, all stores (NL = 4: the 2-deep pipelined size; 16, 32: phases)
template <int NL, bool BLOCKSYNC>
__global__ __launch_bounds__(256) void k_phase(f4v* a, size_t n4) {
  const size_t tile = (size_t)64 * NL;
  const size_t ntiles = n4 / tile;
  const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
  const int lane = threadIdx.x & 63;
  for (size_t i = gw; i < ntiles; i += nw) {
    f4v* p = a + i * tile + lane;
    f4v v[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) v[k] = __builtin_nontemporal_load(p + 64 * k);
    if (BLOCKSYNC) __syncthreads();  // the whole block turns from reading to writing together
#pragma unroll
    for (int k = 0; k < NL; ++k) __builtin_nontemporal_store(v[k] * 0.5f, p + 64 * k);
  }
}

// ---- M: rows of 4 KiB; of every `period` rows the first `active` are read-modify-written, the rest zero-filled.
// One block per `period` rows (a frame), wave w takes rows w, w+4, ...; SPLIT: active rows and zero rows in two kernels.
template <int MODE>  // 0 both, 1 only the RMW rows, 2 only the zero rows
__global__ __launch_bounds__(256) void k_mix(f4v* a, int period, int active) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f4v* blk = a + (size_t)blockIdx.x * period * 256;
  for (int r = wv; r < period; r += 4) {
    f4v* p = blk + (size_t)r * 256 + lane;
    if (r < active) {
      if (MODE == 2) continue;
      f4v v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = __builtin_nontemporal_load(p + 64 * k);
#pragma unroll
      for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(v[k] * 0.5f, p + 64 * k);
    } else {
      if (MODE == 1) continue;
      const f4v z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(z, p + 64 * k);
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
  printf("== W: write-only fill of %.2f GB (best of 6)\n", gb);
  { float ms = timeit([&] { CK(hipMemsetAsync(a, 0, nfloat * 4)); }); printf("%-40s %8.3f ms %7.1f GB/s\n", "hipMemsetAsync", ms, gb / ms * 1e3); }
#define FILL(W, NT, UNR, BLK) { float ms = timeit([&] { k_fill<W, NT, UNR><<<BLK, 256>>>(a, nfloat); }); \
    printf("width %2d %-5s unroll %d blocks %6d        %8.3f ms %7.1f GB/s\n", W, NT ? "nt" : "plain", UNR, BLK, ms, gb / ms * 1e3); }
  for (int blk : {512, 1024, 2048, 4096, 16384}) { FILL(16, true, 4, blk); FILL(16, false, 4, blk); }
  FILL(16, true, 1, 2048); FILL(16, false, 1, 2048); FILL(16, true, 8, 2048); FILL(16, false, 8, 2048);
  FILL(8, true, 4, 2048); FILL(8, false, 4, 2048); FILL(4, true, 4, 2048); FILL(4, false, 4, 2048);
  FILL(4, false, 8, 4096); FILL(8, false, 8, 4096);
  printf("== P: in-place RMW of %.2f GB, reads and writes of a wave / block separated (GB/s counts read + write)\n", gb);
#define PHASE(NL, BS, BLK) { float ms = timeit([&] { k_phase<NL, BS><<<BLK, 256>>>((f4v*)a, n4); }); \
    printf("tile %2d KiB/wave %-10s blocks %6d      %8.3f ms %7.1f GB/s\n", NL, BS ? "block-sync" : "wave", BLK, ms, 2 * gb / ms * 1e3); }
  for (int blk : {512, 1024, 2048}) { PHASE(4, false, blk); PHASE(8, false, blk); PHASE(16, false, blk); PHASE(32, false, blk);
                                      PHASE(8, true, blk); PHASE(16, true, blk); PHASE(32, true, blk); }
  printf("== M: frames of 101 rows x 4 KiB, first `active` rows RMW, rest zero-filled (time; GB/s counts all bytes moved)\n");
  const int period = 101; const int nblk = (int)(n4 / (256 * period));
  for (int active : {0, 24, 44, 101}) {
    const double bytes = (double)nblk * (period * 4096.0 + active * 4096.0);
    float m0 = timeit([&] { k_mix<0><<<nblk, 256>>>((f4v*)a, period, active); });
    float m1 = active ? timeit([&] { k_mix<1><<<nblk, 256>>>((f4v*)a, period, active); }) : 0.f;
    float m2 = active < period ? timeit([&] { k_mix<2><<<nblk, 256>>>((f4v*)a, period, active); }) : 0.f;
    printf("active %3d/101: fused %7.3f ms %7.1f GB/s | RMW rows alone %7.3f ms, zero rows alone %7.3f ms, sum %7.3f ms\n", active, m0,
           bytes / m0 / 1e6, m1, m2, m1 + m2);
  }
  return 0;
}
