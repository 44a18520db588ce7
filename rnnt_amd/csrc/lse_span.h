// Part of the single translation unit rnnt_mi355x.hip: k_lse_span, the first pass for 16-byte aligned rows of up to
// 4096 logits.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "workspace.h"
#include "device_util.h"
#include "grad_span.h"

namespace {

// ---------------------------------------------------------------------------
// k_lse_span: same arithmetic as k_lse (lse_rows.h) in the block shape of k_grad_span: one block per run of
// kSpanWaves * NG * R consecutive rows of an utterance, every wave owning NG * R consecutive rows (about 32 KiB)
// whose loads it issues all at once; rows that hold no lattice cell are not read.  The block's records are
// staged in LDS and leave as one coalesced burst.
// ---------------------------------------------------------------------------
#ifndef RNNT_LSE_WAVES
#define RNNT_LSE_WAVES 8
#endif
#ifndef RNNT_LSE_NG  // row groups per wave for rows of up to 4 KiB (a multiple of 4)
#define RNNT_LSE_NG 4
#endif
constexpr int kLseWaves = RNNT_LSE_WAVES;

template <int L, int NC, int NG>
__global__ __launch_bounds__(kLseWaves * 64) void k_lse_span(const float* __restrict__ x,
                                                             const int32_t* __restrict__ labels,
                                                             const int32_t* __restrict__ in_len,
                                                             const int32_t* __restrict__ lab_len, SpanGeom gm, Ws w) {
  constexpr int R = 64 / L;
  constexpr int RW = NG * R;
  constexpr int RB = kLseWaves * RW;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane % L, r = lane / L;
  const int bl = blockIdx.x / gm.nchunks;
  const int b = gm.b0 + bl;
  const int row0 = (blockIdx.x - bl * gm.nchunks) * RB;
  const int V = gm.V, blank = gm.blank, maxU1 = gm.maxU1;
  const int nchunk = V >> 2;
  const int rows_blk = min(RB, gm.rows_utt - row0);
  const float* xblk = x + (size_t)b * gm.xstride + (size_t)row0 * V;
  int T, U;
  utt_dims(in_len, lab_len, b, gm.maxT, maxU1, T, U);

  __shared__ float4 recs[RB];
  float4* frame0 = w.rec + ((size_t)b * (gm.maxT + 2) + 1) * maxU1;  // frame 0 of this utterance
  const float4 neg = make_float4(kNegF, kNegF, 0.f, 0.f);

  // Blocks beyond the utterance's last frame (padding of a ragged batch) hold no cell: their records are "log 0",
  // and only those of the first U + 18 frames past the end are ever looked at (the sweeps follow anti-diagonals up
  // to T - 1 + U and request records up to 16 diagonals ahead).
  if (row0 >= T * maxU1) {
    if (row0 < (T + U + 18) * maxU1 || T == 0)
      for (int i = threadIdx.x; i < rows_blk; i += kLseWaves * 64) frame0[row0 + i] = neg;
    if (row0 == 0)
      for (int i = threadIdx.x; i < maxU1; i += kLseWaves * 64) (frame0 - maxU1)[i] = neg;
    if (row0 + RB >= gm.rows_utt)
      for (int i = threadIdx.x; i < maxU1; i += kLseWaves * 64) (frame0 + (size_t)gm.rows_utt)[i] = neg;
    return;
  }

  // Every load of the block is issued before anything waits: the target symbols of the wave's rows first (their
  // value is only needed once the rows have arrived: the label's and the blank's logit are picked out of the row's
  // registers, not loaded on their own -- no address depends on a load, and no one-lane loads share the rows' path:
  // 1.58 -> 1.56 ms at B=64, T=400, U=100, V=1024), then the rows themselves.
  float v[NG][NC][4];
  float xb[NG];
  int lab[NG];
  int state[NG];  // 0: no cell, 1: cell without a label move, 2: cell with one
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int rho = wv * RW + g * R + r;
    const int row = row0 + rho;
    const int t = row / maxU1;
    const int u = row - t * maxU1;
    const bool valid = rho < rows_blk && t < T && u <= U;
    state[g] = valid ? (u < U ? 2 : 1) : 0;
    lab[g] = 0;
    if (state[g] == 2) lab[g] = labels[(size_t)b * (maxU1 - 1) + u];
  }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int rho = wv * RW + g * R + r;
    const float* rp = xblk + (size_t)rho * V;
    xb[g] = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = j + L * i;
      if (state[g] && c < nchunk) {
        load_vec<4>(rp + (size_t)c * 4, v[g][i]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[g][i][q] = -INFINITY;
      }
    }
  }

#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int rho = wv * RW + g * R + r;
    int lb = lab[g];
    if (lb < 0 || lb >= V) {
      atomicOr(&w.hdr->flags, RNNT_MI355X_FLAG_BAD_LABEL);
      lb = 0;
    }
    // The label's and the blank's logit out of the row's registers (no one-lane loads next to the row loads).
    float mx = -INFINITY, xl;
    if constexpr (NC < 16) {  // a compare and a select per element and symbol: only the lane that holds it picks it
      float cl = 0.f, cb = 0.f;
#pragma unroll
      for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          mx = fmaxf(mx, v[g][i][q]);
          cl = ((j + L * i) * 4 + q == lb) ? v[g][i][q] : cl;
          cb = ((j + L * i) * 4 + q == blank) ? v[g][i][q] : cb;
        }
      mx = group_max<L>(mx);
      xl = __shfl(cl, r * L + ((lb >> 2) & (L - 1)), 64);
      xb[g] = __shfl(cb, r * L + ((blank >> 2) & (L - 1)), 64);  // (every lane of the group gets it; lane j == 0 uses it)
    } else {
      // Rows of more than 8 KiB (64 registers per lane): a wavefront's 64 lanes share one row, so the symbol is
      // wave-uniform -- element e sits in lane (e >> 2) & 63, register (e >> 8) * 4 + (e & 3): a tree of NC * 4 - 1
      // selects on the bits of the register's number, their conditions scalar, and one readlane per symbol instead of
      // 2 * NC * 4 compares and selects (B=32, T=250, U=150, V=4096: first pass 2.90 -> 2.84 ms; with 16 registers per
      // lane, V=1024, the tree's serial scalar steps cost more than they save: 1.57 -> 1.59 ms).
#pragma unroll
      for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) mx = fmaxf(mx, v[g][i][q]);
      mx = group_max<L>(mx);
      static_assert(NC < 16 || L == 64, "one row per wavefront instruction");
      auto pick = [&](int e) {  // e wave-uniform, 0 <= e < V
        const int eu = __builtin_amdgcn_readfirstlane(e);
        // (bit-field inserts under a scalar mask: written as selects, or as an indexed array, the compiler puts the
        // row's registers into scratch memory to index them)
        unsigned t[NC * 4];
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) t[i * 4 + q] = __float_as_uint(v[g][i][q]);
        unsigned k = (unsigned)(((eu >> 8) << 2) | (eu & 3));
#pragma unroll
        for (int n = NC * 4; n > 1; n >>= 1) {
          const unsigned odd = 0u - (k & 1u);  // all ones: take the odd one of each pair
#pragma unroll
          for (int m = 0; m < n / 2; ++m) asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(t[m]) : "s"(odd), "v"(t[2 * m + 1]), "v"(t[2 * m]));
          k >>= 1;
        }
        return __int_as_float(__builtin_amdgcn_readlane((int)t[0], (eu >> 2) & 63));
      };
      xl = pick(lb);
      xb[g] = pick(blank);
    }
    const float m2 = (state[g] ? mx : 0.f) * kLog2e;  // rows without a cell keep m = 0 so that nothing below is NaN
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) s += fast_exp2(__builtin_fmaf(v[g][i][q], kLog2e, -m2));
    s = group_sum<L>(s);
    // log-probabilities as (x*log2e - m2) - log2(s) with the very m2 the sum was built with: their error then scales
    // with the distance from the row maximum, not with the size of the logits
    const float ls = fast_log2(s);
    if (j == 0) {
      float4 rc = make_float4(kNegF, kNegF, 0.f, 0.f);
      if (state[g])
        rc = make_float4(__builtin_fmaf(xb[g], kLog2e, -m2) - ls,
                         state[g] == 2 ? __builtin_fmaf(xl, kLog2e, -m2) - ls : kNegF, m2 + ls, -ls);
      recs[rho] = rc;
    }
  }
  __syncthreads();
  float4* out = frame0 + row0;
  for (int i = threadIdx.x; i < rows_blk; i += kLseWaves * 64) out[i] = recs[i];
  // the first and the last block of an utterance also write its guard frames
  if (row0 == 0)
    for (int i = threadIdx.x; i < maxU1; i += kLseWaves * 64) (frame0 - maxU1)[i] = neg;
  if (row0 + RB >= gm.rows_utt)
    for (int i = threadIdx.x; i < maxU1; i += kLseWaves * 64) (frame0 + (size_t)gm.rows_utt)[i] = neg;
}

}  // namespace
