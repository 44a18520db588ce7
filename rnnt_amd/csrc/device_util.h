// Part of the single translation unit rnnt_mi355x.hip: device helpers shared by the kernels (exp2/log2, vector loads, DPP reductions, the row cursor).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "workspace.h"

namespace {

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

// Lengths of utterance b, or T = 0 (no valid cell) when they are out of range.
__device__ __forceinline__ void utt_dims(const int32_t* in_len, const int32_t* lab_len, int b, int maxT, int maxU1,
                                         int& T, int& U) {
  const int t = in_len[b], u = lab_len[b];
  const bool ok = (t >= 1) & (t <= maxT) & (u >= 0) & (u <= maxU1 - 1);
  T = ok ? t : 0;
  U = ok ? u : 0;
}

// The logits are streamed exactly once per kernel: non-temporal loads and
// stores (measured on MI355X: 7.0 vs 6.3 TB/s for a pure read stream).
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    const f4v q = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  } else if constexpr (VEC == 2) {
    const f2v q = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(p));
    v[0] = q.x; v[1] = q.y;
  } else {
    v[0] = __builtin_nontemporal_load(p);
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    f4v q = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(q, reinterpret_cast<f4v*>(p));
  } else if constexpr (VEC == 2) {
    f2v q = {v[0], v[1]};
    __builtin_nontemporal_store(q, reinterpret_cast<f2v*>(p));
  } else {
    __builtin_nontemporal_store(v[0], p);
  }
}

// Plain (cached, write-back) forms of the same accesses.
template <int VEC>
__device__ __forceinline__ void load_vec_plain(const float* p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    const f4v q = *reinterpret_cast<const f4v*>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  } else if constexpr (VEC == 2) {
    const f2v q = *reinterpret_cast<const f2v*>(p);
    v[0] = q.x; v[1] = q.y;
  } else {
    v[0] = *p;
  }
}
template <int VEC>
__device__ __forceinline__ void store_vec_plain(float* p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    f4v q = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f4v*>(p) = q;
  } else if constexpr (VEC == 2) {
    f2v q = {v[0], v[1]};
    *reinterpret_cast<f2v*>(p) = q;
  } else {
    *p = v[0];
  }
}

// All-reduce inside aligned groups of L lanes.  The steps inside a row of 16
// lanes are DPP operand modifiers (quad swaps, half-row mirror, row mirror:
// one VALU instruction each, no LDS); only the 16- and 32-lane steps go
// through the LDS crossbar.  max and + are commutative, so paired lanes end up
// with bit-identical values.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int L, typename Op>
__device__ __forceinline__ float group_reduce(float v, Op op) {
  static_assert(L == 8 || L == 16 || L == 32 || L == 64, "group size");

  v = op(v, dpp_mov<0xB1>(v));   // quad_perm [1,0,3,2]
  v = op(v, dpp_mov<0x4E>(v));   // quad_perm [2,3,0,1]
  v = op(v, dpp_mov<0x141>(v));  // row_half_mirror
  if constexpr (L >= 16) v = op(v, dpp_mov<0x140>(v));  // row_mirror
  if constexpr (L >= 32) v = op(v, __shfl_xor(v, 16, 64));
  if constexpr (L >= 64) v = op(v, __shfl_xor(v, 32, 64));
  return v;
}
template <int L>
__device__ __forceinline__ float group_max(float v) {
  return group_reduce<L>(v, [](float a, float b) { return fmaxf(a, b); });
}
template <int L>
__device__ __forceinline__ float group_sum(float v) {
  return group_reduce<L>(v, [](float a, float b) { return a + b; });
}

// Rows of the current tile owned by this lane, advanced incrementally (no
// division in the loop).  Row rho of the block is frame t0 + tt, column uu.
template <int G>
struct RowCursor {
  int rho[G], tt[G], uu[G];
  int pass;
  __device__ __forceinline__ void init(int first_row, int rows_apart, int maxU1) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      rho[g] = first_row + g * rows_apart;
      tt[g] = rho[g] / maxU1;
      uu[g] = rho[g] - tt[g] * maxU1;
    }
    pass = 0;
  }
  __device__ __forceinline__ void advance(int npass, int S, int sq, int sr, int maxU1) {
    if (++pass == npass) {
      pass = 0;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        rho[g] += S;
        tt[g] += sq;
        uu[g] += sr;
        if (uu[g] >= maxU1) {
          uu[g] -= maxU1;
          tt[g] += 1;
        }
      }
    }
  }
};

}  // namespace
