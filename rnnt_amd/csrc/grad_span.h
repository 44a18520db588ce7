// Part of the single translation unit rnnt_mi355x.hip: k_grad_span, the gradient pass for 16-byte aligned rows of up
// to 4096 logits.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "workspace.h"
#include "device_util.h"
#include "cell_coef.h"

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
// So: a span is a run of WAVES * NG * R consecutive rows of an utterance.  A block first turns alpha, beta,
// the likelihood and the records of its rows into {c2, pb, pl, label} in LDS (cell_coef.h: rows that hold no lattice
// cell or whose elements are all negligible are marked inactive), then every wave issues ALL loads of its active rows
// (NG * NC float4 per lane), the block meets at a barrier, and every wave computes and stores its rows (zeros for
// inactive ones, plus the blank's and the label's entry of sparse ones).  A row is owned by L lanes (R = 64 / L rows
// side by side in a wave), lane j holding the 16-byte chunks j, j + L, ... (NC of them: rows of up to 4 * L * NC
// floats).  A block takes NS consecutive spans one after the other: the coefficients of all of them come out of ONE
// round trip to memory (the prologue has a thread per row to spare), so the latency ahead of a block's first row load
// is paid once per NS spans.
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
#ifndef RNNT_SPAN_NS  // spans a block takes one after the other (fewer where the coefficient table would outgrow kSpanMaxRows)
#define RNNT_SPAN_NS 1
#endif
#ifndef RNNT_SPAN_OCC  // waves per SIMD the register allocation must leave room for (two resident blocks of 6 waves)
#define RNNT_SPAN_OCC 3
#endif
constexpr int kSpanWaves = RNNT_SPAN_WAVES;
constexpr int kSpanMaxRows = 512;  // rows per block, upper bound over the instantiations

// rows of a span and spans per block of the instantiation <L, NC, NG>
template <int L, int NG>
struct SpanShape {
  static constexpr int R = 64 / L;            // rows side by side in a wave
  static constexpr int RW = NG * R;           // rows per wave and span
  static constexpr int RB = kSpanWaves * RW;  // rows per span
  static constexpr int NS = RNNT_SPAN_NS * RB <= kSpanMaxRows ? RNNT_SPAN_NS : (kSpanMaxRows / RB > 0 ? kSpanMaxRows / RB : 1);
  static constexpr int ROWS = RB * NS;        // rows per block
};

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
__global__ __launch_bounds__(kSpanWaves * 64) __attribute__((amdgpu_waves_per_eu(RNNT_SPAN_OCC))) void k_grad_span(const float* x, float* grad,
                                                              const int32_t* __restrict__ labels,
                                                              const int32_t* __restrict__ in_len,
                                                              const int32_t* __restrict__ lab_len,
                                                              const float* __restrict__ grad_scale, SpanGeom gm, Ws w) {
  using Shape = SpanShape<L, NG>;
  constexpr int R = Shape::R, RW = Shape::RW, RB = Shape::RB, NS = Shape::NS, ROWS = Shape::ROWS;
  static_assert(ROWS <= kSpanMaxRows, "coefficient table");
  static_assert(RB >= kMinSpanRows, "the workspace holds one partial-sum slot per kMinSpanRows rows");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane % L, r = lane / L;
  const int bid = gm.nblocks - 1 - (int)blockIdx.x;  // reverse of the first pass: most recently read rows first
  const int bl = bid / gm.nchunks;
  const int b = gm.b0 + bl;
  const int row0 = (bid - bl * gm.nchunks) * ROWS;   // first row (t * maxU1 + u) of this block
  const int V = gm.V, blank = gm.blank, maxU1 = gm.maxU1;
  const int nchunk = V >> 2;
  const int rows_blk = min(ROWS, gm.rows_utt - row0);
  const float* xblk = x + (size_t)b * gm.xstride + (size_t)row0 * V;
  float* gblk = grad + (size_t)b * gm.gstride + (size_t)row0 * V;
  const float sc = grad_scale ? grad_scale[b] : 1.f;
  int T, U;
  utt_dims(in_len, lab_len, b, gm.maxT, maxU1, T, U);

  __shared__ float4 coef_s[ROWS];
  __shared__ int cnt_s[2][kSpanWaves];
  __shared__ float red[kSpanWaves];
  {
    int n_cell = 0, n_pruned = 0;  // wave-uniform (ballot counts)
    const int U1s = w.U1s;
    const size_t plane = (size_t)b * w.D * U1s;
    const double ll2 = w.ll[b];
    for (int i = threadIdx.x; i < ROWS; i += kSpanWaves * 64) {
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
        CellRaw raw;
        raw.a = w.alpha[ci]; raw.be = w.beta[ci]; raw.be_t = w.beta[ci_t]; raw.be_u = w.beta[ci_u];
        raw.ll = ll2;
        raw.rc = w.rec[((size_t)b * (gm.maxT + 2) + 1 + t) * maxU1 + u];
        int lab = 0;
        if (maxU1 > 1) lab = labels[(size_t)b * (maxU1 - 1) + min(u, maxU1 - 2)];
        raw.lab = (lab < 0 || lab >= V) ? 0 : lab;
        raw.flags = cell_flags(is_cell, t, u, T, U);
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
    for (int i = 0; i < kSpanWaves; ++i) { c.x += cnt_s[0][i]; c.y += cnt_s[1][i]; }
    w.pcount[gm.b0 * gm.nchunks + bid] = c;
  }

  float sqsum = 0.f;
#pragma unroll 1
  for (int s = 0; s < NS; ++s) {
    const int span0 = s * RB;  // first row of this span inside the block
    if (s > 0 && span0 >= rows_blk) break;  // (block-uniform; a block holds at least one row)
    // ---- read phase: every load of this wave's active rows
    float v[NG][NC][4];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int rho = span0 + wv * RW + g * R + r;  // row inside the block; a wave's rows are consecutive
      const bool active = rho < rows_blk && coef_s[rho < ROWS ? rho : 0].x > -INFINITY;
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
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      int rho = span0 + wv * RW + g * R + r;
      // (the coefficients are looked up again, not kept from the read phase: hidden from common-subexpression
      // elimination, which would otherwise hold four more registers per row group across the barrier)
      asm volatile("" : "+v"(rho));
      if (rho >= rows_blk) continue;
      const float4 cf = coef_s[rho];
      const float c2 = cf.x, pb = cf.y, pl = cf.z;
      const int lab = __float_as_int(cf.w);
      const bool sparse = (pb != 0.f) | (pl != 0.f);  // (only looked at when c2 = -inf)
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
          } else if (sparse) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float gv = unread_elem(c * 4 + q, blank, lab, pb, pl, sc);
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
