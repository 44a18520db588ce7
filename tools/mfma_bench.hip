// Peak rate of v_mfma_f32_32x32x2_f32 on this device: every wave runs a chain of MFMAs on NACC
// independent accumulators, with 1, 2 or 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
// RANDOM: operands differ per lane and change every iteration (data toggling costs power, and the clock follows)
template <int NACC, bool RANDOM>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  if (RANDOM) {
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    a = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
    b = (float)((h * 40503u) >> 8) * (1.0f / 16777216.0f) - 0.5f;
  }
  f16v acc[NACC];
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    if (RANDOM) {
      a = -a * 0.999f;
      b = b * 1.001f - 0.01f * a;
    }
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, bool RANDOM>
void run(int blocks_per_cu) {
  const int blocks = 256 * blocks_per_cu, iters = 65536;
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  (void)hipGetLastError(); k<NACC, RANDOM><<<blocks, 256>>>(out, iters, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  (void)hipGetLastError(); k<NACC, RANDOM><<<blocks, 256>>>(out, iters, 1.f, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * NACC * 4096.0;
  printf("%s operands, NACC=%d waves/SIMD=%d: %.3f ms, %.1f TFLOP/s, %.1f cycles@2.4GHz per MFMA per SIMD\n", RANDOM ? "varying " : "constant", NACC, blocks_per_cu, ms,
         flops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * NACC * blocks_per_cu));
  hipFree(out);
}
int main() {
  run<2, false>(1); run<4, false>(2); run<2, true>(1); run<4, true>(2); run<4, true>(2); run<2, true>(4);
  return 0;
}
