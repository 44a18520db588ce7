// Streaming-bandwidth calibration for the row kernels' access shapes (MI355X).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/membench tools/membench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ inline float4 nt_load(const float4* p) { f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ inline void nt_store(float4 o, float4* p) { f4v v = {o.x, o.y, o.z, o.w}; __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(p)); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE 0: read-only sum; 1: read + exp2 sum; 2: copy to dst; 3: in-place exp2; 4: in-place scale; 5: write-only
template <int MODE, bool NT>
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ src, float4* dst, float* sink, size_t n4, int rows_per_block) {
  // block handles a contiguous span of rows_per_block KiB-rows (each 256 float4 = 4 KiB); wave per row
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t base = (size_t)blockIdx.x * rows_per_block * 256;
  float acc = 0.f;
  for (int r = wv; r < rows_per_block; r += 4) {
    const size_t o = base + (size_t)r * 256;
    if (o + 256 > n4) break;
    float4 v[4];
    if (MODE != 5) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = NT ? nt_load(&src[o + lane + 64 * i]) : src[o + lane + 64 * i];
    }
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc += __builtin_amdgcn_exp2f(v[i].x) + __builtin_amdgcn_exp2f(v[i].y) + __builtin_amdgcn_exp2f(v[i].z) + __builtin_amdgcn_exp2f(v[i].w);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float4 o4;
        if (MODE == 2) o4 = v[i];
        else if (MODE == 3) o4 = make_float4(__builtin_amdgcn_exp2f(v[i].x), __builtin_amdgcn_exp2f(v[i].y), __builtin_amdgcn_exp2f(v[i].z), __builtin_amdgcn_exp2f(v[i].w));
        else if (MODE == 4) o4 = make_float4(v[i].x * 0.5f, v[i].y * 0.5f, v[i].z * 0.5f, v[i].w * 0.5f);
        else o4 = make_float4(1.f, 2.f, 3.f, 4.f);
        if (NT) nt_store(o4, &dst[o + lane + 64 * i]); else dst[o + lane + 64 * i] = o4;
      }
    }
  }
  if (MODE <= 1 && acc == 12345.678f) sink[0] = acc;
}

template <int MODE, bool NT>
double run(const char* name, const float4* src, float4* dst, float* sink, size_t n4, int rpb, double bytes, int reps = 10) {
  const int blocks = (int)((n4 / 256 + rpb - 1) / rpb);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) k_stream<MODE, NT><<<blocks, 256>>>(src, dst, sink, n4, rpb);
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) k_stream<MODE, NT><<<blocks, 256>>>(src, dst, sink, n4, rpb);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
  printf("%-34s rpb=%4d blocks=%7d  %8.3f ms  %7.1f GB/s\n", name, rpb, blocks, ms, bytes / ms / 1e6);
  return ms;
}

int main(int argc, char** argv) {
  const size_t gb = argc > 1 ? atol(argv[1]) : 10;
  const size_t n4 = gb * (1ull << 30) / 16;
  float4 *a, *b; float* sink;
  CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16)); CK(hipMalloc(&sink, 256));
  CK(hipMemset(a, 0x3c, n4 * 16)); CK(hipMemset(b, 0, n4 * 16));
  const double rb = (double)n4 * 16;
  for (int rpb : {101, 404, 1616}) {
    run<0, false>("read sum", a, b, sink, n4, rpb, rb);
    run<0, true>("read sum (nt)", a, b, sink, n4, rpb, rb);
    run<1, false>("read exp2 sum", a, b, sink, n4, rpb, rb);
    run<2, false>("copy a->b", a, b, sink, n4, rpb, 2 * rb);
    run<2, true>("copy a->b (nt)", a, b, sink, n4, rpb, 2 * rb);
    run<4, false>("in-place scale", a, a, sink, n4, rpb, 2 * rb);
    run<4, true>("in-place scale (nt)", a, a, sink, n4, rpb, 2 * rb);
    run<3, false>("in-place exp2", b, b, sink, n4, rpb, 2 * rb);
    run<5, false>("write only", a, b, sink, n4, rpb, rb);
    run<5, true>("write only (nt)", a, b, sink, n4, rpb, rb);
  }
  // hipMemcpy D2D for comparison
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipMemcpyAsync(b, a, n4 * 16, hipMemcpyDeviceToDevice, 0));
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) CK(hipMemcpyAsync(b, a, n4 * 16, hipMemcpyDeviceToDevice, 0));
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  printf("%-34s %8.3f ms  %7.1f GB/s\n", "hipMemcpy D2D", ms, 2 * rb / ms / 1e6);
  return 0;
}
