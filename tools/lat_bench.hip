// Latency and issue cost of the vector operations the lattice sweep is made of, on one wavefront (the sweep is one
// wavefront per utterance and direction).  "dep": a chain of dependent operations of one kind; "ind": eight
// independent chains interleaved (cost per operation when the pipeline is kept full).
//   hipcc --offload-arch=gfx950 -O3 -o tools/lat_bench tools/lat_bench.hip && tools/lat_bench
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kIters = 1 << 18;

template <int OP>
__device__ __forceinline__ void op(double& x, float& f, int& lo, double seed, float fseed) {
  if (OP == 0) x = x + seed;                                                     // v_add_f64
  if (OP == 1) x = fmax(x, seed) ;                                               // v_max_f64
  if (OP == 2) {                                                                 // v_cvt_f32_f64 + v_cvt_f64_f32
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(x));
    asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x) : "v"(f));
  }
  if (OP == 7) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x) : "v"(f));         // v_cvt_f64_f32 alone (f never changes)
  if (OP == 8) asm volatile("v_max_f64 %0, %1, %2" : "=v"(x) : "v"(x), "v"(seed));
  if (OP == 3) f = __builtin_amdgcn_exp2f(-f);                                   // v_exp_f32
  if (OP == 4) f = __builtin_amdgcn_logf(f);                                     // v_log_f32
  if (OP == 5) f = f + fseed;                                                    // v_add_f32
  if (OP == 6) lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false); // v_mov_b32_dpp wave_shr:1
}

template <int OP, int NCH>
__global__ __launch_bounds__(64) void k_chain(double* out, double seed, float fseed) {
  double x[NCH];
  float f[NCH];
  int lo[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) { x[c] = seed + threadIdx.x + c; f[c] = fseed + threadIdx.x * 1e-3f + c; lo[c] = threadIdx.x + c; }
#pragma unroll 1
  for (int i = 0; i < kIters; i += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < NCH; ++c) op<OP>(x[c], f[c], lo[c], seed, fseed);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < NCH; ++c) s += x[c] + f[c] + lo[c];
  out[threadIdx.x] = s;
}

template <int OP, int NCH>
float run(double* d) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  k_chain<OP, NCH><<<1, 64>>>(d, 1.0, 0.5f);
  (void)hipEventRecord(a);
  k_chain<OP, NCH><<<1, 64>>>(d, 1.0, 0.5f);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1e6f / kIters / NCH;  // ns per operation
}

int main() {
  double* d; (void)hipMalloc(&d, 64 * sizeof(double));
  printf("%-28s dep %6.2f ns   ind %6.2f ns\n", "v_add_f64", run<0, 1>(d), run<0, 8>(d));
  printf("%-28s dep %6.2f ns   ind %6.2f ns\n", "v_max_f64", run<1, 1>(d), run<1, 8>(d));
  printf("%-28s dep %6.2f ns   ind %6.2f ns  (two operations)\n", "cvt f64->f32->f64", run<2, 1>(d), run<2, 8>(d));
  printf("%-28s                 ind %6.2f ns\n", "v_cvt_f64_f32", run<7, 8>(d));
  printf("%-28s dep %6.2f ns   ind %6.2f ns\n", "v_max_f64 (asm)", run<8, 1>(d), run<8, 8>(d));
  printf("%-28s dep %6.2f ns   ind %6.2f ns\n", "v_exp_f32", run<3, 1>(d), run<3, 8>(d));
  printf("%-28s dep %6.2f ns   ind %6.2f ns\n", "v_log_f32", run<4, 1>(d), run<4, 8>(d));
  printf("%-28s dep %6.2f ns   ind %6.2f ns\n", "v_add_f32", run<5, 1>(d), run<5, 8>(d));
  printf("%-28s dep %6.2f ns   ind %6.2f ns\n", "v_mov_b32_dpp wave_shr:1", run<6, 1>(d), run<6, 8>(d));
  return 0;
}
