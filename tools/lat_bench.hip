// Latency of dependent VALU operations on one wavefront (the lattice sweep is one long dependent chain per
// wavefront): N iterations of a chain of a single operation kind, timed with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lat_bench tools/lat_bench.hip && tools/lat_bench
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kIters = 1 << 20;

template <int OP>
__global__ __launch_bounds__(64) void k_chain(double* out, double seed, float fseed) {
  double x = seed + threadIdx.x;
  float f = fseed + threadIdx.x * 1e-3f;
  int lo = threadIdx.x;
#pragma unroll 1
  for (int i = 0; i < kIters; i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (OP == 0) x = x + seed;                                                   // v_add_f64
      if (OP == 1) x = fmax(x * 1.0000001, seed);                                  // v_mul_f64 + v_max_f64
      if (OP == 2) x = (double)((float)x) + 0.0;                                   // cvt f64->f32, cvt f32->f64 (+ add folded?)
      if (OP == 3) f = __builtin_amdgcn_exp2f(-f);                                 // v_exp_f32
      if (OP == 4) f = __builtin_amdgcn_logf(f + 2.0f);                            // v_add_f32 + v_log_f32
      if (OP == 5) f = f + fseed;                                                  // v_add_f32
      if (OP == 6) lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);  // v_mov_b32_dpp wave_shr:1
      if (OP == 7) lo = __builtin_amdgcn_update_dpp(lo, lo, 0x111, 0xf, 0xf, false);  // v_mov_b32_dpp row_shr:1
      if (OP == 8) f = (float)((double)f);                                         // cvt f32->f64, cvt f64->f32
    }
  }
  out[threadIdx.x] = x + f + lo;
}

template <int OP>
float run(double* d) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k_chain<OP><<<1, 64>>>(d, 1.0, 0.5f);
  hipEventRecord(a);
  k_chain<OP><<<1, 64>>>(d, 1.0, 0.5f);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e6f / kIters;  // ns per iteration
}

int main() {
  double* d; hipMalloc(&d, 64 * sizeof(double));
  printf("add_f64            %.2f ns\n", run<0>(d));
  printf("mul_f64+max_f64    %.2f ns\n", run<1>(d));
  printf("cvt64->32->64(+add) %.2f ns\n", run<2>(d));
  printf("exp_f32            %.2f ns\n", run<3>(d));
  printf("add_f32+log_f32    %.2f ns\n", run<4>(d));
  printf("add_f32            %.2f ns\n", run<5>(d));
  printf("dpp wave_shr:1     %.2f ns\n", run<6>(d));
  printf("dpp row_shr:1      %.2f ns\n", run<7>(d));
  printf("cvt32->64->32      %.2f ns\n", run<8>(d));
  return 0;
}
