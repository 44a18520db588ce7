// Part of the single translation unit rnnt_mi355x.hip: k_grad_span, the gradient pass for 16-byte aligned rows of up
// to 4096 logits.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "workspace.h"
#include "device_util.h"

namespace {

// ---------------------------------------------------------------------------
// k_grad_span: same arithmetic as k_grad (grad_rows.h), different traffic shape.
//
// In-place read-modify-write traffic is fastest on this memory system when a large contiguous span turns from being
// read to being written at once, and when every wavefront owns a contiguous piece of it: measured on MI355X with
// 4 KiB rows of which 44 of every 101 are rewritten and the rest zero-filled (the mix this pass sees), waves that
// interleave rows 4 KiB at a time and pipeline load/store per row move 5.1 TB/s, one-shot blocks of 6-8 waves that
// each load 8 consecutive rows (32 KiB), meet at a barrier and then store them move 6.6 TB/s.
//
// So: one block per run of WAVES * NG * R consecutive rows of an utterance.  The block first turns alpha, beta,
// the likelihood and the records of its rows into {c2, pb, pl, label} in LDS (rows that hold no lattice cell or
// whose occupancy is negligible are marked inactive), then every wave issues ALL loads of its active rows
// (NG * NC float4 per lane), the block meets at a barrier, and every wave computes and stores its rows (zeros for
// inactive ones).  A row is owned by L lanes (R = 64 / L rows side by side in a wave), lane j holding the 16-byte
// chunks j, j + L, ... (NC of them: rows of up to 4 * L * NC floats).
// ---------------------------------------------------------------------------
#ifndef RNNT_SPAN_WAVES
#define RNNT_SPAN_WAVES 6
#endif
#ifndef RNNT_SPAN_NG  // row groups per wave for rows of up to 4 KiB (a multiple of 4)
#define RNNT_SPAN_NG 8
#endif
#ifndef RNNT_SPAN_PACE  // stores (of 1 KiB per wavefront) between two drains of the write phase; 0 = leave it to the compiler
#define RNNT_SPAN_PACE 4
#endif
constexpr int kSpanWaves = RNNT_SPAN_WAVES;
constexpr int kSpanMaxRows = 512;  // rows per block, upper bound over the instantiations

struct SpanGeom {
  int maxT, maxU1, V, blank;
  int rows_utt;     // maxT * maxU1
  int nchunks;      // blocks per utterance = ceil(rows_utt / rows per block)
  int b0;           // first utterance of this launch
  int nblocks;      // blocks of this launch
  int nblocks_all;  // blocks of the whole batch
  float prune_log2;
  size_t xstride, gstride;
};

template <int L, int NC, int NG>
__global__ __launch_bounds__(kSpanWaves * 64) void k_grad_span(const float* x, float* grad,
                                                              const int32_t* __restrict__ labels,
                                                              const int32_t* __restrict__ in_len,
                                                              const int32_t* __restrict__ lab_len,
                                                              const float* __restrict__ grad_scale, SpanGeom gm, Ws w) {
  constexpr int R = 64 / L;            // rows side by side in a wave
  constexpr int RW = NG * R;           // rows per wave
  constexpr int RB = kSpanWaves * RW;  // rows per block
  static_assert(RB <= kSpanMaxRows, "coefficient table");
  static_assert(RB >= kMinSpanRows, "the workspace holds one partial-sum slot per kMinSpanRows rows");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane % L, r = lane / L;
  const int bid = gm.nblocks - 1 - (int)blockIdx.x;  // reverse of the first pass: most recently read rows first
  const int bl = bid / gm.nchunks;
  const int b = gm.b0 + bl;
  const int row0 = (bid - bl * gm.nchunks) * RB;     // first row (t * maxU1 + u) of this block
  const int V = gm.V, blank = gm.blank, maxU1 = gm.maxU1;
  const int nchunk = V >> 2;
  const int rows_blk = min(RB, gm.rows_utt - row0);
  const float* xblk = x + (size_t)b * gm.xstride + (size_t)row0 * V;
  float* gblk = grad + (size_t)b * gm.gstride + (size_t)row0 * V;
  const float sc = grad_scale ? grad_scale[b] : 1.f;
  int T, U;
  utt_dims(in_len, lab_len, b, gm.maxT, maxU1, T, U);

  __shared__ float4 coef_s[RB];
  __shared__ int cnt_s[2][kSpanWaves];
  __shared__ float red[kSpanWaves];
  {
    int n_cell = 0, n_pruned = 0;  // wave-uniform (ballot counts)
    const int U1s = w.U1s;
    const size_t plane = (size_t)b * w.D * U1s;
    const double ll2 = w.ll[b];
    for (int i = threadIdx.x; i < RB; i += kSpanWaves * 64) {
      const int row = row0 + i;
      const int t = row / maxU1;
      const int u = row - t * maxU1;
      float4 cf = make_float4(-INFINITY, 0.f, 0.f, __int_as_float(-1));
      const bool in_tensor = i < rows_blk;
      const bool is_cell = in_tensor && t < T && u <= U;
#ifdef RNNT_SPAN_FAKE_PROLOGUE  // timing experiment only (wrong results): no dependent loads before the row loads
      if (is_cell && u < 44) cf = make_float4(-20.f, 0.f, 0.f, __int_as_float(-1));
      if (false) {
#else
      if (in_tensor) {
#endif
        // Everything a cell needs is requested at once, for every row of the tensor and before the utterance's
        // lengths are looked at (addresses clamped into the arrays; what is loaded for a row without a cell is not
        // used): one round trip to memory instead of three dependent ones ahead of the block's row loads.
        const size_t ci = plane + (size_t)(t + u) * U1s + u;
        const size_t ci_t = (t + u + 1 < w.D) ? ci + U1s : ci;
        const size_t ci_u = (u + 1 < U1s) ? ci_t + 1 : ci_t;
        const double al = w.alpha[ci], be = w.beta[ci], be_t = w.beta[ci_t], be_u = w.beta[ci_u];
        const float4 rc = w.rec[((size_t)b * (gm.maxT + 2) + 1 + t) * maxU1 + u];  // {lpb, lpl, lse, -}
        int lab = 0;
        if (maxU1 > 1) lab = labels[(size_t)b * (maxU1 - 1) + min(u, maxU1 - 2)];
        const double a_ll = al - ll2;
        const float g2 = (float)(a_ll + be);  // log2 gamma
        if (is_cell && g2 >= gm.prune_log2) {
          float e_b = -INFINITY;  // 2^-inf = 0: no blank move out of this cell
          if (t < T - 1) e_b = (float)(a_ll + be_t) + rc.x;
          else if (u == U) e_b = (float)a_ll + rc.x;
          float plv = 0.f;
          if (u < U) {
            lab = (lab < 0 || lab >= V) ? 0 : lab;
            plv = fast_exp2((float)(a_ll + be_u) + rc.y);
          } else {
            lab = -1;
          }
          cf = make_float4(g2 - rc.z, fast_exp2(e_b), plv, __int_as_float(lab));
        }
      }
      coef_s[i] = cf;
      n_cell += __popcll(__ballot(is_cell));
      n_pruned += __popcll(__ballot(is_cell && !(cf.x > -INFINITY)));
    }
    if (lane == 0) { cnt_s[0][wv] = n_cell; cnt_s[1][wv] = n_pruned; }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) w.hdr->n_partial = gm.nblocks_all;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint2 c = make_uint2(0u, 0u);
#pragma unroll
    for (int i = 0; i < kSpanWaves; ++i) { c.x += cnt_s[0][i]; c.y += cnt_s[1][i]; }
    w.pcount[gm.b0 * gm.nchunks + bid] = c;
  }

  // ---- read phase: every load of this wave's active rows
  float v[NG][NC][4];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int rho = wv * RW + g * R + r;  // row inside the block; a wave's rows are consecutive
    const bool active = rho < rows_blk && coef_s[rho < RB ? rho : 0].x > -INFINITY;
    const float* rp = xblk + (size_t)rho * V;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = j + L * i;
      if (active && c < nchunk) {
        load_vec<4>(rp + (size_t)c * 4, v[g][i]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[g][i][q] = 0.f;
      }
    }
  }
  __syncthreads();  // the whole span turns from reading to writing
#if RNNT_SPAN_PACE > 0
  // The write phase is paced on purpose.  How many stores a wavefront has in flight matters on this memory system:
  // measured at V = 1024 on N(0,1) / peaked logits, a drain after every 1 / 2 / 4 / 8 / 16 / 32 stores of 1 KiB gives
  // 2.85 / 2.49 / 2.31 / 2.31 / 2.35 / 2.40 ms and 2.18 / 1.90 / 1.81 / 1.87 / 1.90 / 1.94 ms.  Every row register is
  // made "used" here, so that the ONE wait for the (conditional, hence uncountable) row loads stands in straight-line
  // code ahead of the first store; left to itself the compiler puts an s_waitcnt vmcnt(0) in front of every row
  // group's arithmetic, which paces the stores by accident (4 per drain at V = 1024: 2.35 / 1.81 ms; 16 at V = 4096).
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(v[g][i][q]));
#endif

  // ---- write phase
  float sqsum = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int rho = wv * RW + g * R + r;
    if (rho >= rows_blk) continue;
    const float4 cf = coef_s[rho];
    const float c2 = cf.x, pb = cf.y, pl = cf.z;
    const int lab = __float_as_int(cf.w);
    float* gp = gblk + (size_t)rho * V;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = j + L * i;
      if (c < nchunk) {
        float o[4];
        if (c2 > -INFINITY) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int e = c * 4 + q;
            float gv = fast_exp2(__builtin_fmaf(v[g][i][q], kLog2e, c2));
            gv -= (e == blank) ? pb : 0.f;
            gv -= (e == lab) ? pl : 0.f;
            gv *= sc;
            sqsum = __builtin_fmaf(gv, gv, sqsum);
            o[q] = gv;
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = 0.f;
        }
        store_vec<4>(gp + (size_t)c * 4, o);
      }
#if RNNT_SPAN_PACE > 0
      if ((g * NC + i + 1) % RNNT_SPAN_PACE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    }
  }

  // per-block sum of squared gradient elements (one slot per block: deterministic)
  sqsum = group_sum<64>(sqsum);
  if (lane == 0) red[wv] = sqsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kSpanWaves; ++i) s += red[i];
    w.partial[gm.b0 * gm.nchunks + bid] = s;
  }
}

}  // namespace
