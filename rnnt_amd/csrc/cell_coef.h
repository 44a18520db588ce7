// Part of the single translation unit rnnt_mi355x.hip: what the gradient of one row needs about its lattice cell.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "workspace.h"
#include "device_util.h"

namespace {

// ---------------------------------------------------------------------------
// The gradient of the row of cell (t, u) with respect to its logits x is
//   dx[v] = s * ( 2^(x[v]*log2e + c2) - [v == blank] * pb - [v == label] * pl )
//   c2 = log2 gamma - lse2,  gamma = P(the path visits (t,u) | labels) = 2^(alpha + beta - ll)
//   pb = P(blank move out of (t,u) | labels),  pl = P(label move out of (t,u) | labels)
// and every gradient kernel first folds alpha, beta, the likelihood and the cell's record into the four numbers
// {c2, pb, pl, label}.  Three kinds of rows come out of that (T = 2^prune_log2, p_max = largest probability of the row,
// which the first pass leaves in the record):
//   full    gamma * p_max >= T          the row's logits are read a second time, every element is computed
//   sparse  gamma * p_max <  T <= gamma every element except the blank's and the label's is gamma * p(v) < T: written
//                                       as exact zeros, logits not read; the two others are written exactly,
//                                       s * (gamma*p(v) - pb) and s * (gamma*p(v) - pl), from the record alone.
//                                       Encoded as c2 = -inf with pb, pl replaced by those differences, so that the
//                                       formula above with "2^-inf = 0" still holds.
//   pruned  gamma < T                   every element is below T in magnitude: exact zeros, c2 = -inf, pb = pl = 0
// Whatever is written as zero without being computed is therefore smaller than grad_scale * T in magnitude.
// Diffuse posteriors (random-init networks, large vocabularies) have p_max << 1 and many sparse rows; peaked ones
// have p_max ~ 1 and the criterion falls back to gamma < T.
// ---------------------------------------------------------------------------

#ifndef RNNT_SPARSE_ROWS  // 0: experiment builds without sparse rows (full / pruned on gamma alone everywhere)
#define RNNT_SPARSE_ROWS 1
#endif
constexpr bool kSparseRows = RNNT_SPARSE_ROWS != 0;

// Everything a cell's coefficients are computed from, as loaded (no arithmetic yet, so that the loads can be issued
// ahead of their use).
struct CellRaw {
  double a, be, be_t, be_u, ll;  // alpha(t,u), beta(t,u), beta(t+1,u), beta(t,u+1), log2-likelihood
  float4 rc;                     // {log2 P(blank), log2 P(label), log2-sum-exp2, log2 p_max}
  int lab;                       // target symbol of the label move, clamped into [0,V); -1 without one
  int flags;                     // kCell*
};
constexpr int kCellValid = 1;      // (t,u) is a lattice cell
constexpr int kCellBlankNext = 2;  // t < T-1: the blank move stays inside the lattice
constexpr int kCellLastCol = 4;    // u == U (the blank move out of (T-1,U) ends the path)
constexpr int kCellHasLabel = 8;   // u < U

__device__ __forceinline__ int cell_flags(bool is_cell, int t, int u, int T, int U) {
  return is_cell ? (kCellValid | (t < T - 1 ? kCellBlankNext : 0) | (u == U ? kCellLastCol : 0) |
                    (u < U ? kCellHasLabel : 0))
                 : 0;
}

// {c2, pb, pl, label}.  SPARSE = false keeps the two-way split (full / pruned on gamma alone) for the kernels whose
// zero rows never pass through the element loop.
template <bool SPARSE>
__device__ __forceinline__ float4 cell_coef(const CellRaw& r, float prune_log2, int blank) {
  float4 cf = make_float4(-INFINITY, 0.f, 0.f, __int_as_float(-1));
  if (r.flags & kCellValid) {
    const double a_ll = r.a - r.ll;
    const float g2 = (float)(a_ll + r.be);  // log2 gamma
    if (g2 >= prune_log2) {
      float e_b = -INFINITY;  // 2^-inf = 0: no blank move out of this cell
      if (r.flags & kCellBlankNext) e_b = (float)(a_ll + r.be_t) + r.rc.x;
      else if (r.flags & kCellLastCol) e_b = (float)a_ll + r.rc.x;
      const bool has_lab = r.flags & kCellHasLabel;
      const float pbv = fast_exp2(e_b);
      const float plv = has_lab ? fast_exp2((float)(a_ll + r.be_u) + r.rc.y) : 0.f;
      const int lab = has_lab ? r.lab : -1;
      if (!SPARSE || g2 + r.rc.w >= prune_log2) {
        cf = make_float4(g2 - r.rc.z, pbv, plv, __int_as_float(lab));
      } else {
        // (a label that equals the blank shares its element: gamma * p is then subtracted once)
        const float gpb = fast_exp2(g2 + r.rc.x);
        const float gpl = (has_lab && lab != blank) ? fast_exp2(g2 + r.rc.y) : 0.f;
        cf = make_float4(-INFINITY, pbv - gpb, plv - gpl, __int_as_float(lab));
      }
    }
  }
  return cf;
}

// One element of a row whose logits were not read (c2 = -inf): zero, or the blank's / label's exact entry of a
// sparse row.
__device__ __forceinline__ float unread_elem(int e, int blank, int lab, float pb, float pl, float sc) {
  float gv = 0.f;
  gv -= (e == blank) ? pb : 0.f;
  gv -= (e == lab) ? pl : 0.f;
  return gv * sc;
}

}  // namespace
