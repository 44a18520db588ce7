// Part of the single translation unit rnnt_mi355x.hip: k_lse, first streaming pass over the logits.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "workspace.h"
#include "device_util.h"

namespace {

// ---------------------------------------------------------------------------
// k_lse: row log-sum-exp + blank/label log-probs.
//
// One block per (utterance, TT frames): the block's rows are contiguous in
// memory and share T_b/U_b.  A wavefront is split into R = 64/L groups of L
// lanes; each group owns one row per tile, lane j of the group owns the
// VEC-wide chunks j, j+L, j+2L, ...  A tile is G*R rows x NC chunks per lane
// (G*NC = 4 vector loads in flight per lane); tiles are software-pipelined two
// deep so that a wave always has the next tile's loads outstanding while it
// reduces the current one.
// ---------------------------------------------------------------------------
template <int VEC, int NC, int G>
struct LseTile {
  float v[G][NC][VEC];
  float xb[G], xl[G];  // lane j==0, pass 0: logits of blank and of the next label
  int rho[G];          // row inside the block, -1 = the row holds no cell
  int haslab[G];
  int pass;
};

template <int VEC, int L, int NC, int G>
__global__ __launch_bounds__(kRowBlock) void k_lse(const float* __restrict__ x, const int32_t* __restrict__ labels,
                                                   const int32_t* __restrict__ in_len,
                                                   const int32_t* __restrict__ lab_len, RowGeom gm, Ws w) {
  constexpr int R = 64 / L;
  constexpr int S = kRowWaves * G * R;  // rows per block iteration
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane % L, r = lane / L;
  const int bl = blockIdx.x / gm.nTB;  // utterance within this launch
  const int b = gm.b0 + bl;
  const int t0 = (blockIdx.x - bl * gm.nTB) * gm.TT;
  int T, U;
  utt_dims(in_len, lab_len, b, gm.maxT, gm.maxU1, T, U);
  // frames at or beyond T_b hold no lattice cell: their logits are not read, their records are "log 0"
  const int rows_cell = (max(min(t0 + gm.TT, T), t0) - t0) * gm.maxU1;
  const int rows_blk = (min(t0 + gm.TT, gm.maxT) - t0) * gm.maxU1;
  const int ntiles = ((rows_cell + S - 1) / S) * gm.npass;
  const int V = gm.V, blank = gm.blank, maxU1 = gm.maxU1, npass = gm.npass;
  const int nchunk = V / VEC;
  const int sq = S / maxU1, sr = S % maxU1;
  const float* xblk = x + (size_t)b * gm.xstride + (size_t)t0 * maxU1 * V;
  extern __shared__ float4 recs[];  // this block's records, written out in one coalesced burst at the end

  for (int i = threadIdx.x; i < rows_blk; i += kRowBlock) recs[i] = make_float4(kNegF, kNegF, 0.f, 0.f);
  __syncthreads();

  RowCursor<G> cur;
  cur.init(wv * G * R + r, R, maxU1);

  float m_run[G], s_run[G], exb[G], exl[G];
#pragma unroll
  for (int g = 0; g < G; ++g) { m_run[g] = -INFINITY; s_run[g] = 0.f; exb[g] = 0.f; exl[g] = 0.f; }

  auto issue = [&](LseTile<VEC, NC, G>& tl) {
    tl.pass = cur.pass;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const bool valid = (cur.rho[g] < rows_cell) & (cur.uu[g] <= U);
      tl.rho[g] = valid ? cur.rho[g] : -1;
      tl.haslab[g] = valid & (cur.uu[g] < U);
      const float* rp = xblk + (size_t)(valid ? cur.rho[g] : 0) * V;
      tl.xb[g] = 0.f;
      tl.xl[g] = 0.f;
      if (cur.pass == 0 && valid && j == 0) {
        tl.xb[g] = rp[blank];
        if (cur.uu[g] < U) {
          int lab = labels[(size_t)b * (maxU1 - 1) + cur.uu[g]];
          if (lab < 0 || lab >= V) {
            atomicOr(&w.hdr->flags, RNNT_MI355X_FLAG_BAD_LABEL);
            lab = 0;
          }
          tl.xl[g] = rp[lab];
        }
      }
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int c = j + L * (cur.pass * NC + i);
        if (valid && c < nchunk) {
          load_vec<VEC>(rp + (size_t)c * VEC, tl.v[g][i]);
        } else {
#pragma unroll
          for (int q = 0; q < VEC; ++q) tl.v[g][i][q] = -INFINITY;
        }
      }
    }
    cur.advance(npass, S, sq, sr, maxU1);
  };

  auto consume = [&](const LseTile<VEC, NC, G>& tl) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (tl.pass == 0) {
        m_run[g] = -INFINITY;
        s_run[g] = 0.f;
        exb[g] = tl.xb[g];
        exl[g] = tl.xl[g];
      }
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int q = 0; q < VEC; ++q) mx = fmaxf(mx, tl.v[g][i][q]);
      mx = group_max<L>(mx);
      // rows without a cell keep m = 0 so that nothing below is NaN
      const float m_new = (tl.rho[g] >= 0) ? fmaxf(m_run[g], mx) : 0.f;
      const float m2 = m_new * kLog2e;
      float s = s_run[g] * fast_exp2(m_run[g] * kLog2e - m2);
#pragma unroll
      for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int q = 0; q < VEC; ++q) s += fast_exp2(__builtin_fmaf(tl.v[g][i][q], kLog2e, -m2));
      m_run[g] = m_new;
      s_run[g] = s;
    }
    if (tl.pass == npass - 1) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float s_all = group_sum<L>(s_run[g]);
        // log-probabilities as (x*log2e - m2) - log2(s) with the very m2 the sum was built with: their error then
        // scales with the distance from the row maximum, not with the size of the logits (m2 + log2(s) rounded
        // first would cost an ulp of the log-sum-exp, 8e-6 at logits near 100)
        const float m2 = m_run[g] * kLog2e, ls = fast_log2(s_all);
        if (tl.rho[g] >= 0 && j == 0)
          recs[tl.rho[g]] = make_float4(__builtin_fmaf(exb[g], kLog2e, -m2) - ls,
                                        tl.haslab[g] ? __builtin_fmaf(exl[g], kLog2e, -m2) - ls : kNegF, m2 + ls, -ls);
      }
    }
  };

  LseTile<VEC, NC, G> ta, tb;
  issue(ta);
  for (int s = 0; s < ntiles; s += 2) {
    if (s + 1 < ntiles) issue(tb);
    consume(ta);
    if (s + 1 >= ntiles) break;
    if (s + 2 < ntiles) issue(ta);
    consume(tb);
  }
  __syncthreads();
  float4* frame0 = w.rec + ((size_t)b * (gm.maxT + 2) + 1) * maxU1;  // frame 0 of this utterance
  float4* out = frame0 + (size_t)t0 * maxU1;
  for (int i = threadIdx.x; i < rows_blk; i += kRowBlock) out[i] = recs[i];
  // the first and the last block of an utterance also write its guard frames
  const float4 neg = make_float4(kNegF, kNegF, 0.f, 0.f);
  if (t0 == 0)
    for (int i = threadIdx.x; i < maxU1; i += kRowBlock) (frame0 - maxU1)[i] = neg;
  if (t0 + gm.TT >= gm.maxT)
    for (int i = threadIdx.x; i < maxU1; i += kRowBlock) (frame0 + (size_t)gm.maxT * maxU1)[i] = neg;
}

}  // namespace
