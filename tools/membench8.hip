// Zero-fill of the rows of a [B][rows][row_kib KiB] tensor whose bit in a row mask is clear (the gradient pass's
// k_fill_rows), on MI355X: what an empty wavefront costs, what block shape and what footprint do to the rate.
// usage: membench8 [GB of tensor, default 10.6] [row KiB, default 4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));

// A: one wavefront per KiB, WAVES per block (k_fill_rows as built)
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_a(float* g, long long kib_total, int row_kib, const uint32_t* __restrict__ bits) {
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long kib = (long long)blockIdx.x * WAVES + wv;
  if (kib >= kib_total) return;
  const long long row = kib / row_kib;
  const uint32_t word = bits[row >> 5];
  if ((word >> (row & 31)) & 1u) return;
  const f4v z = {0.f, 0.f, 0.f, 0.f};
  __builtin_nontemporal_store(z, reinterpret_cast<f4v*>(g + kib * 256) + (threadIdx.x & 63));
}
// B: one wavefront per row (row_kib stores), WAVES per block
template <int WAVES, int RK>
__global__ __launch_bounds__(WAVES * 64) void k_b(float* g, long long rows, const uint32_t* __restrict__ bits) {
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long row = (long long)blockIdx.x * WAVES + wv;
  if (row >= rows) return;
  const uint32_t word = bits[row >> 5];
  if ((word >> (row & 31)) & 1u) return;
  const f4v z = {0.f, 0.f, 0.f, 0.f};
  f4v* p = reinterpret_cast<f4v*>(g + row * 256 * RK) + (threadIdx.x & 63);
#pragma unroll
  for (int k = 0; k < RK; ++k) __builtin_nontemporal_store(z, p + 64 * k);
}
// C: no mask at all (pure fill, 1 KiB per wavefront)
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_c(float* g, long long kib_total) {
  const long long kib = (long long)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (kib >= kib_total) return;
  const f4v z = {0.f, 0.f, 0.f, 0.f};
  __builtin_nontemporal_store(z, reinterpret_cast<f4v*>(g + kib * 256) + (threadIdx.x & 63));
}
// D: through a list of the rows to fill (no empty wavefront but those past the list's end); wave per KiB
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_d(float* g, const int* __restrict__ list, const int* __restrict__ count, int row_kib) {
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long j = (long long)blockIdx.x * WAVES + wv;
  const long long e = j / row_kib;
  if (e >= *count) return;
  const long long kib = (long long)list[e] * row_kib + (j - e * row_kib);
  const f4v z = {0.f, 0.f, 0.f, 0.f};
  __builtin_nontemporal_store(z, reinterpret_cast<f4v*>(g + kib * 256) + (threadIdx.x & 63));
}
// E: a block of 16 wavefronts owns 32 rows (one word of the mask); wavefront w takes rows w and w + 16
template <int RK>
__global__ __launch_bounds__(1024) void k_e(float* g, long long rows, const uint32_t* __restrict__ bits) {
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t word = bits[blockIdx.x];
  const f4v z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = wv + 16 * h;
    const long long row = (long long)blockIdx.x * 32 + r;
    if (row < rows && !((word >> r) & 1u)) {
      f4v* p = reinterpret_cast<f4v*>(g + row * 256 * RK) + (threadIdx.x & 63);
#pragma unroll
      for (int k = 0; k < RK; ++k) __builtin_nontemporal_store(z, p + 64 * k);
    }
  }
}

// R: k_fill_rows as the library has it (two-dimensional grid, one utterance per blockIdx.y)
__global__ __launch_bounds__(256) void k_r(float* __restrict__ grad, size_t gstride, int kib_utt, int row_kib,
                                           const uint32_t* __restrict__ rowbits, int row_words) {
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int kib = (int)blockIdx.x * 4 + wv;
  if (kib >= kib_utt) return;
  const int b = blockIdx.y;
  const int row = row_kib == 1 ? kib : kib / row_kib;
  const uint32_t word = rowbits[(size_t)b * row_words + (row >> 5)];
  if ((word >> (row & 31)) & 1u) return;
  const f4v z = {0.f, 0.f, 0.f, 0.f};
  f4v* dst = reinterpret_cast<f4v*>(grad + (size_t)b * gstride + (size_t)kib * 256) + (threadIdx.x & 63);
  __builtin_nontemporal_store(z, dst);
}

template <typename F>
float timeit(F launch, int reps = 5) {
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

int main(int argc, char** argv) {
  const double want_gb = argc > 1 ? atof(argv[1]) : 10.6;
  const int row_kib = argc > 2 ? atoi(argv[2]) : 4;
  const int U1 = 101, T = 400;
  const long long rows_utt = (long long)U1 * T;
  const long long B = (long long)(want_gb * 1e9 / (rows_utt * row_kib * 1024.0));
  const long long rows = B * rows_utt, kib_total = rows * row_kib;
  const double gb = kib_total * 1024.0 / 1e9;
  float* g; CK(hipMalloc(&g, kib_total * 1024)); CK(hipMemset(g, 0x3c, kib_total * 1024));
  uint32_t* bits; CK(hipMalloc(&bits, (rows / 32 + 2 + 64 * B) * 4));
  int* list; CK(hipMalloc(&list, rows * 4));
  int* count; CK(hipMalloc(&count, 4));
  printf("tensor %.2f GB = %lld rows of %d KiB\n", gb, rows, row_kib);
  { float ms = timeit([&] { CK(hipMemsetAsync(g, 0, kib_total * 1024)); }); printf("%-56s %8.3f ms %7.1f GB/s\n", "hipMemsetAsync", ms, gb / ms * 1e3); }
#define C(W) { unsigned n = (unsigned)((kib_total + W - 1) / W); float ms = timeit([&] { k_c<W><<<n, W * 64>>>(g, kib_total); }); \
    printf("no mask, KiB per wave, %2d waves per block                 %8.3f ms %7.1f GB/s\n", W, ms, gb / ms * 1e3); }
  C(1) C(4) C(16)
  {  // the library's kernel on utterances of rows_utt rows, nothing masked
    CK(hipMemset(bits, 0, (rows / 32 + 2 + 64 * B) * 4));
    const int kib_utt = (int)(rows_utt * row_kib), row_words = (int)((rows_utt + 63) / 64 * 2);
    if ((size_t)B * row_words <= (size_t)(rows / 32 + 2 + 64 * B)) {
      float ms = timeit([&] { k_r<<<dim3((unsigned)((kib_utt + 3) / 4), (unsigned)B), 256>>>(g, (size_t)kib_utt * 256, kib_utt, row_kib, bits, row_words); });
      printf("%-56s %8.3f ms %7.1f GB/s\n", "library's k_fill_rows (2-D grid), nothing masked", ms, gb / ms * 1e3);
    }
  }
  for (double band : {0.0, 3.5, 29.0, 101.0}) {  // rows of a frame around the diagonal that are NOT filled
    std::vector<uint32_t> h(rows / 32 + 2, 0u);
    std::vector<int> hl;
    long long nfull = 0;
    for (long long r = 0; r < rows; ++r) {
      const long long ru = r % rows_utt;
      const int t = (int)(ru / U1), u = (int)(ru % U1);
      const bool full = std::fabs(u - t * (double)U1 / T) < band / 2;
      if (full) { h[r >> 5] |= 1u << (r & 31); ++nfull; } else hl.push_back((int)r);
    }
    const int n_zero = (int)hl.size();
    CK(hipMemcpy(bits, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(list, hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(count, &n_zero, 4, hipMemcpyHostToDevice));
    const double zgb = (double)n_zero * row_kib * 1024.0 / 1e9;
    printf("== %.1f %% of the rows filled (%.2f GB)\n", 100.0 * n_zero / rows, zgb);
#define A(W) { unsigned n = (unsigned)((kib_total + W - 1) / W); float ms = timeit([&] { k_a<W><<<n, W * 64>>>(g, kib_total, row_kib, bits); }); \
    printf("A wave per KiB, %2d waves per block                        %8.3f ms %7.1f GB/s\n", W, ms, zgb / ms * 1e3); }
    A(1) A(2) A(4) A(8) A(16)
#define Bk(W, RK) if (row_kib == RK) { unsigned n = (unsigned)((rows + W - 1) / W); float ms = timeit([&] { k_b<W, RK><<<n, W * 64>>>(g, rows, bits); }); \
    printf("B wave per row, %2d waves per block                        %8.3f ms %7.1f GB/s\n", W, ms, zgb / ms * 1e3); }
    Bk(1, 4) Bk(4, 4) Bk(16, 4) Bk(1, 2) Bk(4, 2) Bk(16, 2) Bk(4, 1) Bk(4, 8) Bk(4, 16)
#define D(W) { unsigned n = (unsigned)((kib_total + W - 1) / W); float ms = timeit([&] { k_d<W><<<n, W * 64>>>(g, list, count, row_kib); }); \
    printf("D list of rows, wave per KiB, %2d waves per block (full grid) %6.3f ms %7.1f GB/s\n", W, ms, zgb / ms * 1e3); }
    D(4) D(16)
#define E(RK) if (row_kib == RK) { unsigned n = (unsigned)((rows + 31) / 32); float ms = timeit([&] { k_e<RK><<<n, 1024>>>(g, rows, bits); }); \
    printf("E block of 16 waves per 32 rows                           %8.3f ms %7.1f GB/s\n", ms, zgb / ms * 1e3); }
    E(1) E(2) E(4) E(8) E(16)
  }
  return 0;
}
