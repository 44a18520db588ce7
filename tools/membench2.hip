// In-place read-modify-write streaming variants (MI355X): which shape gets closest to the blit copy rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));

// each wave: NL float4 per lane per iteration (NL KiB), WAVES waves per block, grid-stride over "tiles"
template <int NL, int THREADS, bool NTL, bool NTS, bool CONTIG>
__global__ __launch_bounds__(THREADS) void k_rmw(f4v* a, size_t n4) {
  constexpr int W = THREADS / 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t tile = 64 * NL;  // float4 per wave-iteration
  const size_t ntiles = n4 / tile;
  size_t first, step, count;
  if (CONTIG) {  // each block owns one contiguous span
    const size_t per_blk = (ntiles + gridDim.x - 1) / gridDim.x;
    first = blockIdx.x * per_blk + wv; step = W; count = min(per_blk, ntiles > blockIdx.x * per_blk ? ntiles - blockIdx.x * per_blk : 0);
    for (size_t i = wv; i < count; i += W) {
      f4v* p = a + (blockIdx.x * per_blk + i) * tile + lane;
      f4v v[NL];
#pragma unroll
      for (int k = 0; k < NL; ++k) v[k] = NTL ? __builtin_nontemporal_load(p + 64 * k) : p[64 * k];
#pragma unroll
      for (int k = 0; k < NL; ++k) { f4v o = v[k] * 0.5f; if (NTS) __builtin_nontemporal_store(o, p + 64 * k); else p[64 * k] = o; }
    }
  } else {  // chip-wide interleave: consecutive waves take consecutive tiles
    const size_t gw = (size_t)blockIdx.x * W + wv, nw = (size_t)gridDim.x * W;
    for (size_t i = gw; i < ntiles; i += nw) {
      f4v* p = a + i * tile + lane;
      f4v v[NL];
#pragma unroll
      for (int k = 0; k < NL; ++k) v[k] = NTL ? __builtin_nontemporal_load(p + 64 * k) : p[64 * k];
#pragma unroll
      for (int k = 0; k < NL; ++k) { f4v o = v[k] * 0.5f; if (NTS) __builtin_nontemporal_store(o, p + 64 * k); else p[64 * k] = o; }
    }
  }
  (void)first; (void)step;
}

template <int NL, int THREADS, bool NTL, bool NTS, bool CONTIG>
void run(const char* name, f4v* a, size_t n4, int blocks) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) k_rmw<NL, THREADS, NTL, NTS, CONTIG><<<blocks, THREADS>>>(a, n4);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 8; ++i) k_rmw<NL, THREADS, NTL, NTS, CONTIG><<<blocks, THREADS>>>(a, n4);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 8;
  printf("%-44s blocks=%6d %8.3f ms %7.1f GB/s\n", name, blocks, ms, 2.0 * n4 * 16 / ms / 1e6);
}

int main() {
  const size_t n4 = 10ull * (1ull << 30) / 16;
  f4v* a; CK(hipMalloc(&a, n4 * 16)); CK(hipMemset(a, 0x3c, n4 * 16));
  run<4, 256, false, false, false>("NL4 T256 interleave", a, n4, 2048);
  run<4, 256, true, true, false>("NL4 T256 interleave nt", a, n4, 2048);
  run<8, 256, true, true, false>("NL8 T256 interleave nt", a, n4, 2048);
  run<16, 256, true, true, false>("NL16 T256 interleave nt", a, n4, 1024);
  run<16, 256, false, false, false>("NL16 T256 interleave", a, n4, 1024);
  run<4, 1024, true, true, false>("NL4 T1024 interleave nt", a, n4, 512);
  run<8, 1024, true, true, false>("NL8 T1024 interleave nt", a, n4, 256);
  run<4, 256, true, true, true>("NL4 T256 contig nt 2048blk", a, n4, 2048);
  run<4, 256, true, true, true>("NL4 T256 contig nt 25600blk", a, n4, 25600);
  run<8, 256, true, true, true>("NL8 T256 contig nt 25600blk", a, n4, 25600);
  run<8, 256, true, false, true>("NL8 T256 contig ntL 25600blk", a, n4, 25600);
  run<8, 256, false, true, true>("NL8 T256 contig ntS 25600blk", a, n4, 25600);
  run<16, 256, true, true, true>("NL16 T256 contig nt 25600blk", a, n4, 25600);
  run<4, 256, true, true, true>("NL4 T256 contig nt 102400blk", a, n4, 102400);
  run<4, 64, true, true, true>("NL4 T64 contig nt 102400blk", a, n4, 102400);
  run<4, 256, true, true, false>("NL4 T256 interleave nt 512blk", a, n4, 512);
  run<4, 256, true, true, false>("NL4 T256 interleave nt 1024blk", a, n4, 1024);
  run<2, 256, true, true, false>("NL2 T256 interleave nt 2048blk", a, n4, 2048);
  run<1, 256, true, true, false>("NL1 T256 interleave nt 2048blk", a, n4, 2048);
  return 0;
}
