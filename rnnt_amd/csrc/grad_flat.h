// Part of the single translation unit rnnt_mi355x.hip: k_grad_flat, the gradient pass for utterances that start on
// 16-byte boundaries, with rows of 32 to 4096 logits of any length (no row needs to be a whole number of cache
// lines, or even of 16-byte pieces).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "workspace.h"
#include "device_util.h"
#include "cell_coef.h"

namespace {

// ---------------------------------------------------------------------------
// k_grad_flat: the read-phase / barrier / write-phase shape of k_grad_span (grad_span.h) with the block's span cut
// by position in memory instead of by rows.  The gradient pass needs no reduction along a row, so a row may be
// shared by any lanes: an utterance's [maxT * maxU1, V] logits are taken as a flat run of 16-byte chunks, a block
// owns kFlatWaves * 32 KiB of it, every wave a contiguous 32 KiB, every load / store instruction of a wave one
// contiguous KiB whatever V is (rows of 800 or 1200 bytes straddle cache lines when lanes are tied to rows).
// Each chunk looks up the {c2, pb, pl, label} of its row in the block's LDS table; (row, column) of a lane's
// chunks advance incrementally, without divisions.  When V is no multiple of 4 (the reference's 62-symbol
// vocabulary: 248-byte rows) a chunk may hold the end of one row and the start of the next: it then looks both up.
// ---------------------------------------------------------------------------
#ifndef RNNT_FLAT_WAVES
#define RNNT_FLAT_WAVES 6
#endif
constexpr int kFlatWaves = RNNT_FLAT_WAVES;
constexpr int kFlatNch = 32;                                // 16-byte chunks per lane: 32 KiB per wave
constexpr int kFlatChunksPerBlock = kFlatWaves * 64 * kFlatNch;

struct FlatGeom {
  int maxT, maxU1, V, blank;
  int rows_utt;         // maxT * maxU1
  int q256, r256;       // 256 / V, 256 % V: how (row, column) move from one chunk of a lane to its next
  int blocks_utt;       // blocks per utterance
  int table_rows;       // rows of the LDS coefficient table = 4 * kFlatChunksPerBlock / V + 2
  int b0, nblocks, nblocks_all;
  float prune_log2;
  size_t xstride, gstride;
};

template <bool ROWS4>  // V % 4 == 0: no chunk straddles two rows
__global__ __launch_bounds__(kFlatWaves * 64) __attribute__((amdgpu_waves_per_eu(3))) void k_grad_flat(const float* x, float* grad,
                                                              const int32_t* __restrict__ labels,
                                                              const int32_t* __restrict__ in_len,
                                                              const int32_t* __restrict__ lab_len,
                                                              const float* __restrict__ grad_scale, FlatGeom gm, Ws w) {
  constexpr int NCH = kFlatNch;
  constexpr int CPW = 64 * NCH;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: addresses below stay scalar
  const int bid = gm.nblocks - 1 - (int)blockIdx.x;  // reverse of the first pass: most recently read rows first
  const int bl = bid / gm.blocks_utt;
  const int b = gm.b0 + bl;
  const int kblk = bid - bl * gm.blocks_utt;
  const int V = gm.V, blank = gm.blank, maxU1 = gm.maxU1;
  const long long g0 = (long long)kblk * (4 * kFlatChunksPerBlock);  // first element of this block in the utterance
  const long long total = (long long)gm.rows_utt * V;                // elements of the utterance
  const int row_first = (int)(g0 / V);
  const int rem0 = (int)(g0 - (long long)row_first * V);             // the block starts this far into row_first
  const long long gend = min(g0 + 4 * kFlatChunksPerBlock, total);
  const int nrows = (int)((gend - 1) / V) + 1 - row_first;           // rows this block touches
  const float* xblk = x + (size_t)b * gm.xstride + (size_t)g0;
  float* gblk = grad + (size_t)b * gm.gstride + (size_t)g0;
  const float sc = grad_scale ? grad_scale[b] : 1.f;
  int T, U;
  utt_dims(in_len, lab_len, b, gm.maxT, maxU1, T, U);

  extern __shared__ float4 coef_s[];  // one entry per row the block touches
  __shared__ int cnt_s[2][kFlatWaves];
  __shared__ float red[kFlatWaves];
  {
    int n_cell = 0, n_pruned = 0;  // wave-uniform (ballot counts)
    const int U1s = w.U1s;
    const size_t plane = (size_t)b * w.D * U1s;
    const double ll2 = w.ll[b];
    for (int i = threadIdx.x; i < nrows; i += kFlatWaves * 64) {
      const int row = row_first + i;
      const int t = row / maxU1;
      const int u = row - t * maxU1;
      float4 cf = make_float4(-INFINITY, 0.f, 0.f, __int_as_float(-1));
      const bool is_cell = t < T && u <= U;
      {
        // everything a cell needs, requested at once for every row and before the lengths are looked at (addresses
        // clamped into the arrays; what is loaded for a row without a cell is not used): one round trip
        const size_t ci = plane + (size_t)(t + u) * U1s + u;
        const size_t ci_t = (t + u + 1 < w.D) ? ci + U1s : ci;
        const size_t ci_u = (u + 1 < U1s) ? ci_t + 1 : ci_t;
        CellRaw raw;
        raw.a = w.alpha[ci]; raw.be = w.beta[ci]; raw.be_t = w.beta[ci_t]; raw.be_u = w.beta[ci_u];
        raw.ll = ll2;
        raw.rc = w.rec[((size_t)b * (gm.maxT + 2) + 1 + t) * maxU1 + u];  // {lpb, lpl, lse, log2 p_max}
        int lab = 0;
        if (maxU1 > 1) lab = labels[(size_t)b * (maxU1 - 1) + min(u, maxU1 - 2)];
        raw.lab = (lab < 0 || lab >= V) ? 0 : lab;
        raw.flags = cell_flags(is_cell, t, u, T, U);
        cf = cell_coef<kSparseRows>(raw, gm.prune_log2, blank);
      }
      coef_s[i] = cf;
      // a row that straddles two blocks is counted by the block that holds its first chunk
      const bool mine = is_cell && (i > 0 || rem0 == 0);
      n_cell += __popcll(__ballot(mine));
      n_pruned += __popcll(__ballot(mine && !(cf.x > -INFINITY)));
    }
    if (lane == 0) { cnt_s[0][wv] = n_cell; cnt_s[1][wv] = n_pruned; }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { w.hdr->n_partial = gm.nblocks_all; w.hdr->n_count = gm.nblocks_all; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint2 c = make_uint2(0u, 0u);
#pragma unroll
    for (int i = 0; i < kFlatWaves; ++i) { c.x += cnt_s[0][i]; c.y += cnt_s[1][i]; }
    w.pcount[gm.b0 * gm.blocks_utt + bid] = c;
  }

  // this lane's first chunk: f0 chunks into the block, its first element at (row_l0, col0) in the table's rows
  const int f0 = wv * CPW + lane;
  const int pos0 = rem0 + 4 * f0;
  const int row_l0 = pos0 / V;
  const int col0 = pos0 - row_l0 * V;
  const int left_e = (int)(gend - g0);  // elements of this block inside the utterance
  const int left = left_e >> 2;         // whole chunks (an utterance's last chunk may be partial: handled at the end)
  // a wave's chunk i sits at (scalar base of the wave) + i KiB + 16 * lane: one 32-bit vector offset for all 32
  const char* xw = reinterpret_cast<const char*>(xblk + (size_t)wv * CPW * 4);
  char* gw = reinterpret_cast<char*>(gblk + (size_t)wv * CPW * 4);
  const unsigned voff = 16u * (unsigned)lane;

  // ---- read phase: every load of this wave's chunks that lie in active rows
  float v[NCH][4];
  {
    int row_l = row_l0, col = col0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int f = f0 + 64 * i;
      bool active = f < left && coef_s[f < left ? row_l : 0].x > -INFINITY;
      if (!ROWS4) active = active || (f < left && col + 3 >= V && coef_s[row_l + 1].x > -INFINITY);
      if (active) {
        load_vec<4>(reinterpret_cast<const float*>(xw + 1024 * i + voff), v[i]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[i][q] = 0.f;
      }
      row_l += gm.q256;
      col += gm.r256;
      if (col >= V) { col -= V; ++row_l; }
      // keep the table look-ups next to their loads (hoisting all 32 of them costs registers the rows need)
      if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();  // the whole span turns from reading to writing

  // ---- write phase
  float sqsum = 0.f;
  {
    int row_l = row_l0, col = col0;
    // the walk over (row, column) is redone, not remembered: hidden from common-subexpression elimination, which
    // would otherwise keep all 32 positions of the read phase alive across the barrier (64 registers)
    asm volatile("" : "+v"(row_l), "+v"(col));
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int f = f0 + 64 * i;
      if (f < left) {
        const float4 cf = coef_s[row_l];
        float o[4];
        if (ROWS4 || col + 3 < V) {  // the whole chunk lies in one row
          const float c2 = cf.x, pb = cf.y, pl = cf.z;
          const int lab = __float_as_int(cf.w);
          if (c2 > -INFINITY) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int e = col + q;
              float gv = fast_exp2(__builtin_fmaf(v[i][q], kLog2e, c2));
              gv -= (e == blank) ? pb : 0.f;
              gv -= (e == lab) ? pl : 0.f;
              gv *= sc;
              sqsum = __builtin_fmaf(gv, gv, sqsum);
              o[q] = gv;
            }
          } else if ((pb != 0.f) | (pl != 0.f)) {  // sparse row: only the blank's and the label's entry are not negligible
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float gv = unread_elem(col + q, blank, lab, pb, pl, sc);
              sqsum = __builtin_fmaf(gv, gv, sqsum);
              o[q] = gv;
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = 0.f;
          }
        } else {  // the end of one row and the start of the next
          const float4 cn = coef_s[row_l + 1];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const bool nxt = col + q >= V;
            const int e = nxt ? col + q - V : col + q;
            const float c2 = nxt ? cn.x : cf.x, pb = nxt ? cn.y : cf.y, pl = nxt ? cn.z : cf.z;
            const int lab = __float_as_int(nxt ? cn.w : cf.w);
            float gv = 0.f;
            if (c2 > -INFINITY) {
              gv = fast_exp2(__builtin_fmaf(v[i][q], kLog2e, c2));
              gv -= (e == blank) ? pb : 0.f;
              gv -= (e == lab) ? pl : 0.f;
              gv *= sc;
              sqsum = __builtin_fmaf(gv, gv, sqsum);
            } else if ((pb != 0.f) | (pl != 0.f)) {
              gv = unread_elem(e, blank, lab, pb, pl, sc);
              sqsum = __builtin_fmaf(gv, gv, sqsum);
            }
            o[q] = gv;
          }
        }
        store_vec<4>(reinterpret_cast<float*>(gw + 1024 * i + voff), o);
      }
      row_l += gm.q256;
      col += gm.r256;
      if (col >= V) { col -= V; ++row_l; }
      if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }

  // the utterance's last, partial chunk (its element count is no multiple of 4): element by element
  if (!ROWS4 && (left_e & 3) && threadIdx.x < (left_e & 3)) {
    const int off = (left << 2) + threadIdx.x;  // element offset inside the block
    const int pos = rem0 + off;
    const int rl = pos / V, e = pos - rl * V;
    const float4 cf = coef_s[rl];
    float gv = 0.f;
    if (cf.x > -INFINITY) {
      gv = fast_exp2(__builtin_fmaf(xblk[off], kLog2e, cf.x));
      gv -= (e == blank) ? cf.y : 0.f;
      gv -= (e == __float_as_int(cf.w)) ? cf.z : 0.f;
      gv *= sc;
      sqsum = __builtin_fmaf(gv, gv, sqsum);
    } else if ((cf.y != 0.f) | (cf.z != 0.f)) {
      gv = unread_elem(e, blank, __float_as_int(cf.w), cf.y, cf.z, sc);
      sqsum = __builtin_fmaf(gv, gv, sqsum);
    }
    gblk[off] = gv;
  }

  // per-block sum of squared gradient elements (one slot per block: deterministic)
  sqsum = group_sum<64>(sqsum);
  if (lane == 0) red[wv] = sqsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kFlatWaves; ++i) s += red[i];
    w.partial[gm.b0 * gm.blocks_utt + bid] = s;
  }
}

}  // namespace
