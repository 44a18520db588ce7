// Part of the single translation unit rnnt_mi355x.hip: k_grad, second streaming pass (gradient over the logits).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "workspace.h"
#include "device_util.h"
#include "cell_coef.h"

namespace {

// ---------------------------------------------------------------------------
// k_grad: gradient w.r.t. the logits, written over them (or to `grad`).
//   dx[v] = s * ( 2^(x[v]*log2(e) + c2) - [v==blank] * pb - [v==label] * pl )
//   c2 = log2 gamma - lse,  pb = P(blank move | labels),  pl = P(label move | labels)
// Same block/tile decomposition and two-deep pipeline as k_lse.  A block first
// folds alpha, beta, the likelihood and the records of its cells into
// {c2, pb, pl, label} in LDS (cell_coef.h); rows that hold no lattice cell, or
// whose elements are all negligible, are marked inactive there and are
// zero-filled without being read (sparse rows: plus their two exact entries).  Blocks are taken in the reverse of k_lse's order so
// that the rows k_lse read last (still in the memory-side cache) are re-read
// first.
// ---------------------------------------------------------------------------
// Compile-time tuning knobs of k_grad (experiments build variants of the library with -D...).
#ifndef RNNT_GRAD_ZERO_PLAIN   // zero rows leave through plain stores instead of non-temporal ones
#define RNNT_GRAD_ZERO_PLAIN 0
#endif
#ifndef RNNT_GRAD_STORE_PLAIN  // gradient rows leave through plain stores
#define RNNT_GRAD_STORE_PLAIN 0
#endif
#ifndef RNNT_GRAD_LOAD_PLAIN   // the second read of the logits uses plain loads
#define RNNT_GRAD_LOAD_PLAIN 0
#endif
#ifndef RNNT_GRAD_FWD_ORDER    // blocks in k_lse's order instead of the reverse
#define RNNT_GRAD_FWD_ORDER 0
#endif
#ifndef RNNT_GRAD_WAVES        // waves per SIMD the register allocation aims at
#define RNNT_GRAD_WAVES 8
#endif

template <int VEC, int NC, int G>
struct GradTile {
  float v[G][NC][VEC];
  float4 coef[G];  // {c2, pb, pl, label as int bits}; c2 = -inf marks an inactive row
  int rho[G];      // row inside the block, -1 = beyond the block
  int pass;
};

template <int VEC, int L, int NC, int G>
__global__ __launch_bounds__(kRowBlock) __attribute__((amdgpu_waves_per_eu(RNNT_GRAD_WAVES))) void k_grad(const float* x, float* grad, const int32_t* __restrict__ labels,
                                                    const int32_t* __restrict__ in_len,
                                                    const int32_t* __restrict__ lab_len,
                                                    const float* __restrict__ grad_scale, RowGeom gm, Ws w) {
  constexpr int R = 64 / L;
  constexpr int S = kRowWaves * G * R;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane % L, r = lane / L;
  const int bid = RNNT_GRAD_FWD_ORDER ? (int)blockIdx.x : gm.nblocks - 1 - (int)blockIdx.x;
  const int bl = bid / gm.nTB;  // utterance within this launch
  const int b = gm.b0 + bl;
  const int t0 = (bid - bl * gm.nTB) * gm.TT;
  const int V = gm.V, blank = gm.blank, maxU1 = gm.maxU1, npass = gm.npass;
  const int nframes = min(t0 + gm.TT, gm.maxT) - t0;
  const int rows_blk = nframes * maxU1;  // frames at or beyond T_b are included: they are zero-filled
  const int ntiles = ((rows_blk + S - 1) / S) * npass;
  const int nchunk = V / VEC;
  const int sq = S / maxU1, sr = S % maxU1;
  const float* xblk = x + (size_t)b * gm.xstride + (size_t)t0 * maxU1 * V;
  float* gblk = grad + (size_t)b * gm.gstride + (size_t)t0 * maxU1 * V;
  const float sc = grad_scale ? grad_scale[b] : 1.f;
  int T, U;
  utt_dims(in_len, lab_len, b, gm.maxT, maxU1, T, U);

  extern __shared__ float4 coef_s[];  // one entry per row of the block
  __shared__ int cnt_s[2][kRowWaves];
  {
    int n_cell = 0, n_pruned = 0;  // wave-uniform (ballot counts)
    const int U1s = w.U1s;
    const size_t plane = (size_t)b * w.D * U1s;
    const double ll2 = w.ll[b];
    for (int i = threadIdx.x; i < rows_blk; i += kRowBlock) {
      const int tt = i / maxU1;
      const int u = i - tt * maxU1;
      const int t = t0 + tt;
      float4 cf = make_float4(-INFINITY, 0.f, 0.f, __int_as_float(-1));
      const bool is_cell = t < T && u <= U;
      if (is_cell) {
        const size_t ci = plane + (size_t)(t + u) * U1s + u;
        CellRaw raw;
        raw.a = w.alpha[ci]; raw.be = w.beta[ci]; raw.ll = ll2;
        raw.be_t = (t < T - 1) ? w.beta[ci + U1s] : 0.0;
        raw.rc = w.rec[((size_t)b * (gm.maxT + 2) + 1 + t) * maxU1 + u];  // {lpb, lpl, lse, log2 p_max}
        raw.be_u = 0.0;
        raw.lab = -1;
        if (u < U) {
          raw.be_u = w.beta[ci + U1s + 1];
          const int lab = labels[(size_t)b * (maxU1 - 1) + u];
          raw.lab = (lab < 0 || lab >= V) ? 0 : lab;
        }
        raw.flags = cell_flags(true, t, u, T, U);
        cf = cell_coef<kSparseRows>(raw, gm.prune_log2, blank);
      }
      coef_s[i] = cf;
      n_cell += __popcll(__ballot(is_cell));
      n_pruned += __popcll(__ballot(is_cell && !(cf.x > -INFINITY)));
    }
    if (lane == 0) { cnt_s[0][wv] = n_cell; cnt_s[1][wv] = n_pruned; }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { w.hdr->n_partial = gm.nblocks_all; w.hdr->n_count = gm.nblocks_all; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint2 c = make_uint2(0u, 0u);
#pragma unroll
    for (int i = 0; i < kRowWaves; ++i) { c.x += cnt_s[0][i]; c.y += cnt_s[1][i]; }
    w.pcount[gm.b0 * gm.nTB + bid] = c;
  }

  RowCursor<G> cur;
  cur.init(wv * G * R + r, R, maxU1);

  float sqsum = 0.f;

  auto issue = [&](GradTile<VEC, NC, G>& tl) {
    tl.pass = cur.pass;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const bool in_blk = cur.rho[g] < rows_blk;
      const int rho = in_blk ? cur.rho[g] : 0;
      tl.rho[g] = in_blk ? cur.rho[g] : -1;
      tl.coef[g] = coef_s[rho];
      const bool active = in_blk & (tl.coef[g].x > -INFINITY);
      if (!in_blk) tl.coef[g] = make_float4(-INFINITY, 0.f, 0.f, __int_as_float(-1));
      const float* rp = xblk + (size_t)rho * V;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int c = j + L * (cur.pass * NC + i);
        if (active && c < nchunk) {
          if constexpr (RNNT_GRAD_LOAD_PLAIN) load_vec_plain<VEC>(rp + (size_t)c * VEC, tl.v[g][i]);
          else load_vec<VEC>(rp + (size_t)c * VEC, tl.v[g][i]);
        } else {
#pragma unroll
          for (int q = 0; q < VEC; ++q) tl.v[g][i][q] = 0.f;
        }
      }
    }
    cur.advance(npass, S, sq, sr, maxU1);
  };

  auto consume = [&](const GradTile<VEC, NC, G>& tl) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float c2 = tl.coef[g].x, pb = tl.coef[g].y, pl = tl.coef[g].z;
      const int lab = __float_as_int(tl.coef[g].w);
      float* gp = gblk + (size_t)(tl.rho[g] >= 0 ? tl.rho[g] : 0) * V;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int c = j + L * (tl.pass * NC + i);
        if (tl.rho[g] >= 0 && c < nchunk) {
          float o[VEC];
          if (c2 > -INFINITY) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
              const int e = c * VEC + q;
              float gv = fast_exp2(__builtin_fmaf(tl.v[g][i][q], kLog2e, c2));
              gv -= (e == blank) ? pb : 0.f;
              gv -= (e == lab) ? pl : 0.f;
              gv *= sc;
              sqsum = __builtin_fmaf(gv, gv, sqsum);
              o[q] = gv;
            }
            if constexpr (RNNT_GRAD_STORE_PLAIN) store_vec_plain<VEC>(gp + (size_t)c * VEC, o);
            else store_vec<VEC>(gp + (size_t)c * VEC, o);
          } else {
            if ((pb != 0.f) | (pl != 0.f)) {  // sparse row: only the blank's and the label's entry are not negligible
#pragma unroll
              for (int q = 0; q < VEC; ++q) {
                const float gv = unread_elem(c * VEC + q, blank, lab, pb, pl, sc);
                sqsum = __builtin_fmaf(gv, gv, sqsum);
                o[q] = gv;
              }
            } else {
#pragma unroll
              for (int q = 0; q < VEC; ++q) o[q] = 0.f;
            }
            if constexpr (RNNT_GRAD_ZERO_PLAIN) store_vec_plain<VEC>(gp + (size_t)c * VEC, o);
            else store_vec<VEC>(gp + (size_t)c * VEC, o);
          }
        }
      }
    }
  };

  GradTile<VEC, NC, G> ta, tb;
  issue(ta);
  for (int s = 0; s < ntiles; s += 2) {
    if (s + 1 < ntiles) issue(tb);
    consume(ta);
    if (s + 1 >= ntiles) break;
    if (s + 2 < ntiles) issue(ta);
    consume(tb);
  }

  // per-block sum of squared gradient elements (one slot per block: deterministic)
  __shared__ float red[kRowWaves];
  sqsum = group_sum<64>(sqsum);
  if (lane == 0) red[wv] = sqsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kRowWaves; ++i) s += red[i];
    w.partial[gm.b0 * gm.nTB + bid] = s;
  }
}

}  // namespace
