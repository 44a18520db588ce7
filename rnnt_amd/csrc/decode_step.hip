// Joint + log-softmax of the decoder's inner step for MI355X (gfx950).
//
// The reference's beam search evaluates, for a hypothesis and an encoder frame,
//     out  = pred + encoder_output_t                      (reference rnnt/rnnt_decoder.py:156,170,200)
//     logp = Util.softmax(out, is_log=True)               (reference rnnt/rnnt_decoder.py:158,202;
//                                                          rnnt/base/util.py:295-309: exp(x - max), log, - log sum)
// one V-vector at a time in numpy.  Here any number of (frame, hypothesis) pairs is done in one launch:
//     out[n, :] = log_softmax(f[fi[n], :] + g[gi[n], :])
// with optional row-index arrays (all hypotheses of a frame share its encoder row; NULL = row n) and g
// optional (NULL: the log-softmax of f alone).  One wavefront per output row, the row kept in registers when
// V <= 2048, otherwise re-read; base-2 exponentials as everywhere in this library.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/rnnt_mi355x.h"

namespace {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr int kRegs = 32;  // elements per lane held in registers (V <= 64 * kRegs)

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <bool CACHED>
__global__ __launch_bounds__(256) void k_joint_log_softmax(const float* __restrict__ f, const float* __restrict__ g,
                                                           const int32_t* __restrict__ fi,
                                                           const int32_t* __restrict__ gi, int n_rows, int f_rows,
                                                           int g_rows, int V, float* __restrict__ out,
                                                           int32_t* __restrict__ bad) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= n_rows) return;
  int fr = fi ? fi[n] : n, gr = gi ? gi[n] : n;
  if (fr < 0 || fr >= f_rows || (g && (gr < 0 || gr >= g_rows))) {  // uniform over the wave
    if (lane == 0) atomicOr(bad, 1);
    fr = 0;
    gr = 0;
  }
  const float* fp = f + (size_t)fr * V;
  const float* gp = g ? g + (size_t)gr * V : nullptr;
  float* op = out + (size_t)n * V;
  float x[CACHED ? kRegs : 1];
  float m = -INFINITY;
  if (CACHED) {
#pragma unroll
    for (int k = 0; k < kRegs; ++k) {
      const int v = lane + 64 * k;
      x[k] = -INFINITY;
      if (v < V) x[k] = gp ? fp[v] + gp[v] : fp[v];
      m = fmaxf(m, x[k]);
    }
  } else {
    for (int v = lane; v < V; v += 64) m = fmaxf(m, gp ? fp[v] + gp[v] : fp[v]);
  }
  m = wave_max(m);
  float s = 0.f;
  if (CACHED) {
#pragma unroll
    for (int k = 0; k < kRegs; ++k) s += __builtin_amdgcn_exp2f((x[k] - m) * kLog2e);  // 2^-inf = 0 past the end
  } else {
    for (int v = lane; v < V; v += 64) s += __builtin_amdgcn_exp2f(((gp ? fp[v] + gp[v] : fp[v]) - m) * kLog2e);
  }
  const float lse = m + __builtin_amdgcn_logf(wave_sum(s)) * kLn2;
  if (CACHED) {
#pragma unroll
    for (int k = 0; k < kRegs; ++k) {
      const int v = lane + 64 * k;
      if (v < V) op[v] = x[k] - lse;
    }
  } else {
    for (int v = lane; v < V; v += 64) op[v] = (gp ? fp[v] + gp[v] : fp[v]) - lse;
  }
}

}  // namespace

extern "C" int rnnt_mi355x_joint_log_softmax(const float* f, int f_rows, const float* g, int g_rows,
                                             const int32_t* f_index, const int32_t* g_index, int n_rows, int V,
                                             float* out, int32_t* bad_index_flag, void* hip_stream) {
  if (!f || !out || !bad_index_flag || n_rows < 1 || V < 1 || f_rows < 1 || (g && g_rows < 1))
    return RNNT_MI355X_INVALID_VALUE;
  if (!f_index && f_rows < n_rows) return RNNT_MI355X_INVALID_VALUE;
  if (g && !g_index && g_rows < n_rows) return RNNT_MI355X_INVALID_VALUE;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const unsigned grid = (unsigned)((n_rows + 3) / 4);
  if (V <= 64 * kRegs)
    k_joint_log_softmax<true><<<grid, 256, 0, st>>>(f, g, f_index, g_index, n_rows, f_rows, g_rows, V, out, bad_index_flag);
  else
    k_joint_log_softmax<false><<<grid, 256, 0, st>>>(f, g, f_index, g_index, n_rows, f_rows, g_rows, V, out, bad_index_flag);
  return hipGetLastError() == hipSuccess ? RNNT_MI355X_OK : RNNT_MI355X_LAUNCH_FAILURE;
}
