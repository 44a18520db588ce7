// Part of the single translation unit rnnt_mi355x.hip: the gradient pass split by kind of row (rows of whole KiB).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "workspace.h"
#include "device_util.h"
#include "cell_coef.h"
#include "grad_span.h"

namespace {

// ---------------------------------------------------------------------------
// Most rows of a gradient are exact zeros (cells off the alignment's band, padding of a ragged batch), and the two
// kinds of traffic want different shapes on this memory system.  Zero-fill is fastest as wavefronts that store one
// KiB each and end (measured 6.8-7.3 TB/s; 4 KiB per wavefront 6.2, 16-64 KiB 5.3-5.9, and about 6.0 for the zero
// rows inside k_grad_span's 32-KiB-per-wavefront blocks, which also cannot start before their block's prologue has
// come back from memory).  Read-modify-write wants blocks that load a large piece and turn it from read to written at
// once (grad_span.h) -- and blocks that are FULL of such rows: the span kernel run over the rows that are not zero
// alone, each block skipping the zero rows of its 48 consecutive ones, moves 3.4 TB/s (too few bytes in flight).
// So, for rows of whole KiB, the gradient pass is three kernels:
//   k_classify_rows  one thread per row of the padded tensor, one block per region of kRegionRows consecutive rows of
//                    an utterance: {c2, pb, pl, label} (cell_coef.h); one bit per row into rowbits (set: the row is
//                    full); the coefficients of the full rows, in row order, packed with their row number into the
//                    region's stretch of crec (compaction inside a region needs no pass over the batch and its order
//                    does not depend on timing), those of the sparse rows likewise from the stretch's end; one work
//                    item per rb full rows appended to `items`; the block's counts of lattice cells and of rows that
//                    are not re-read, and the squares of its sparse rows' entries;
//   k_fill_rows      one wavefront per KiB of the tensor: reads its row's bit (a scalar load, served by the scalar
//                    cache: 512 rows per 64-byte line), stores a KiB of zeros if the bit is clear, ends;
//   k_patch_sparse   the two entries of every sparse row, over those zeros (a few 4-byte stores per region);
//   k_grad_rmw       the span kernel's read phase / barrier / write phase over the compacted rows, as a resident
//                    grid whose blocks go through the work items (entries [k * rb, (k+1) * rb) of a region: mostly
//                    runs of consecutive rows, the alignment's band inside one frame after the other) -- a grid with
//                    a block per possible item is mostly blocks that find nothing to do, and each of them holds a
//                    block's registers for a round trip to memory (measured: 2 us, 40 % of the kernel with 70 % of
//                    the rows zero).
// The order of the work items depends on timing (an atomic counter hands out their places); nothing else does: the
// per-block sums of squares are stored by item identity (region, k), not by place, so the batch's sum keeps its fixed
// order and the gradient pass stays bit-reproducible.
// ---------------------------------------------------------------------------
#ifndef RNNT_RMW_PACE  // stores (of 1 KiB per wavefront) between two drains of k_grad_rmw's write phase; 0 = none
#define RNNT_RMW_PACE RNNT_SPAN_PACE
#endif
#ifndef RNNT_RMW_WAVES  // wavefronts per block of k_grad_rmw
#define RNNT_RMW_WAVES RNNT_SPAN_WAVES
#endif
#ifndef RNNT_RMW_NG  // rows per wavefront and work item of k_grad_rmw for rows of up to 4 KiB (a multiple of 4)
#define RNNT_RMW_NG RNNT_SPAN_NG
#endif
#ifndef RNNT_RMW_OCC  // wavefronts per SIMD its register allocation must leave room for
#define RNNT_RMW_OCC RNNT_SPAN_OCC
#endif
constexpr int kRmwWaves = RNNT_RMW_WAVES;
#ifndef RNNT_RMW_WAVES_WIDE  // wavefronts per block for rows of more than 8 KiB (measured at V = 4096: 8 -> 4.00 ms, 6 -> 4.11 ms;
#define RNNT_RMW_WAVES_WIDE 8  // at V = 1024 the other way round: 2.17 against 2.145 ms)
#endif
constexpr int kRmwWavesWide = RNNT_RMW_WAVES_WIDE;
constexpr int kClassifyIters = 4;  // rows per thread of k_classify_rows
constexpr int kClassifyBlock = kRegionRows / kClassifyIters;
static_assert(kRegionRows % (64 * kClassifyIters) == 0 && kClassifyBlock <= 1024 && kRegionRows <= 65536,
              "whole wavefronts; a row's number inside its region is packed into 16 bits");
constexpr int kItemBits = 10;  // a work item's code: (region id << kItemBits) | its number inside the region
constexpr int kItemMask = (1 << kItemBits) - 1;

struct SplitGeom {
  int maxT, maxU1, V, blank;
  int rows_utt;      // maxT * maxU1
  int regions_utt;   // regions per utterance
  int rb;            // entries of a region per work item = rows a block of k_grad_rmw takes at a time
  int spb;           // work items per region at most = ceil(kRegionRows / rb)
  int waves;         // wavefronts per block of k_grad_rmw (rb / waves entries each)
  int nt;            // the lattice state is larger than the memory-side cache: k_classify_rows reads it past the caches
  float prune_log2;
  size_t xstride, gstride;
};

__device__ __forceinline__ float pack_row_label(int local_row, int lab) {
  return __int_as_float((local_row << 16) | (lab & 0xffff));
}

__global__ __launch_bounds__(kClassifyBlock) void k_classify_rows(const int32_t* __restrict__ labels,
                                                                  const int32_t* __restrict__ in_len,
                                                                  const int32_t* __restrict__ lab_len,
                                                                  const float* __restrict__ grad_scale, SplitGeom gm, Ws w) {
  constexpr int NW = kClassifyBlock / 64;
  const int b = blockIdx.y, reg = blockIdx.x;
  const int regid = b * gm.regions_utt + reg;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int maxU1 = gm.maxU1;
  int T, U;
  utt_dims(in_len, lab_len, b, gm.maxT, maxU1, T, U);
  if (reg * kRegionRows >= T * maxU1) {  // (block-uniform) a region of padding: nothing to load, nothing but zeros
    const int row = reg * kRegionRows + kClassifyIters * (int)threadIdx.x;  // the region's rows, 32 per word
    static_assert(kClassifyIters == 4, "a thread in eight clears a word of 32 row bits");
    if ((threadIdx.x & 7) == 0 && (row >> 5) < w.row_words) w.rowbits[(size_t)b * w.row_words + (row >> 5)] = 0u;
    for (int j = threadIdx.x; j <= gm.spb; j += kClassifyBlock) w.partial[(size_t)regid * (gm.spb + 1) + j] = 0.f;
    if (threadIdx.x == 0) {
      const unsigned long long old = atomicAdd(&w.hdr->items_ctr, 1ull << 32);
      if ((int)(old >> 32) == (int)(gridDim.x * gridDim.y) - 1) {
        w.hdr->n_items_ready = (int)(old & 0xffffffffull);
        w.hdr->rmw_next = 0u;
        atomicExch(&w.hdr->items_ctr, 0ull);
      }
      w.rcount[regid] = 0;
      w.pcount[regid] = make_uint2(0u, 0u);
      if (regid == 0) {  // (not reached: region 0 of utterance 0 holds cells unless its lengths are invalid)
        w.hdr->n_count = (int)(gridDim.x * gridDim.y);
        w.hdr->n_partial = (int)(gridDim.x * gridDim.y) * (gm.spb + 1);
      }
    }
    return;
  }
  const int U1s = w.U1s;
  const double ll2 = w.ll[b];
  // Everything the region's cells need is requested at once, before the lengths are looked at (addresses clamped
  // into the arrays; what is loaded for a row without a cell is not used).
  CellRaw raw[kClassifyIters];
  int tt[kClassifyIters], uu[kClassifyIters];
#pragma unroll
  for (int it = 0; it < kClassifyIters; ++it) {
    const int row = reg * kRegionRows + it * kClassifyBlock + (int)threadIdx.x;
    const int rc_row = row < gm.rows_utt ? row : 0;
    const int t = rc_row / maxU1;
    const int u = rc_row - t * maxU1;
    tt[it] = t; uu[it] = u;
    const size_t ci = ((size_t)b * w.D + (t + u)) * U1s + u;
    // Lattice state of more than the memory-side cache holds is read past the caches: lines that ordinary loads
    // leave there are not displaced by the streaming kernels that follow, which then run a quarter slower for
    // as long as they last (measured: the zero-fill of a 52 GB tensor 5.0-5.9 instead of 7.05 TB/s after 1.2 GB of
    // ordinary loads here).  Smaller state stays cached.
    const float4* rcp = &w.rec[((size_t)b * (gm.maxT + 2) + 1 + t) * maxU1 + u];
    raw[it].ll = ll2;
    if (gm.nt) {  // (block-uniform)
      raw[it].a = __builtin_nontemporal_load(&w.alpha[ci]);
      raw[it].be = __builtin_nontemporal_load(&w.beta[ci]);
      const f4v r4 = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(rcp));
      raw[it].rc = make_float4(r4[0], r4[1], r4[2], r4[3]);
    } else {
      raw[it].a = w.alpha[ci]; raw[it].be = w.beta[ci];
      raw[it].rc = *rcp;
    }
    int lab = 0;
    if (maxU1 > 1) lab = labels[(size_t)b * (maxU1 - 1) + min(u, maxU1 - 2)];
    raw[it].lab = (lab < 0 || lab >= gm.V) ? 0 : lab;
  }
  // beta(t+1,u) and beta(t,u+1) are the beta of rows maxU1 and 1 further on: they come from the threads that loaded
  // them, through LDS (alpha and beta are stored by anti-diagonal, so a wavefront's 64 rows are 64 different sectors
  // per load: two such loads per row instead of four), the maxU1 rows behind the region's end from a few more loads.
  // (Used only where the move stays inside the lattice, i.e. where that row is a cell of the same utterance.)
  constexpr int kHalo = (kMaxU1 + kClassifyBlock - 1) / kClassifyBlock;
  __shared__ double be_s[kRegionRows + kHalo * kClassifyBlock];
  double halo[kHalo];
#pragma unroll
  for (int h = 0; h < kHalo; ++h) {
    const int hr = h * kClassifyBlock + (int)threadIdx.x;
    halo[h] = 0.0;
    if (hr < maxU1) {  // (needed rows lie inside the utterance; the others' values are not used)
      const int row = (reg + 1) * kRegionRows + hr;
      const int rc_row = row < gm.rows_utt ? row : 0;
      const int t = rc_row / maxU1, u = rc_row - t * maxU1;
      const double* bp = &w.beta[((size_t)b * w.D + (t + u)) * U1s + u];
      halo[h] = gm.nt ? __builtin_nontemporal_load(bp) : *bp;
    }
  }
#pragma unroll
  for (int it = 0; it < kClassifyIters; ++it) be_s[it * kClassifyBlock + (int)threadIdx.x] = raw[it].be;
#pragma unroll
  for (int h = 0; h < kHalo; ++h) be_s[kRegionRows + h * kClassifyBlock + (int)threadIdx.x] = halo[h];
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kClassifyIters; ++it) {
    const int local_row = it * kClassifyBlock + (int)threadIdx.x;
    raw[it].be_t = be_s[local_row + maxU1];
    raw[it].be_u = be_s[local_row + 1];
  }
  __shared__ int wcnt[2][kClassifyIters][NW];  // full rows / sparse rows per (iteration, wavefront)
  __shared__ int cnt_s[2][NW];
  __shared__ float sq_s[NW];
  const float sc = grad_scale ? grad_scale[b] : 1.f;
  float4 cf[kClassifyIters];
  unsigned long long mask[kClassifyIters], smask[kClassifyIters];
  int n_cell = 0, n_unread = 0;  // wave-uniform (ballot counts)
  float sq = 0.f;                // squares of the sparse rows' two entries, as k_patch_sparse stores them
#pragma unroll
  for (int it = 0; it < kClassifyIters; ++it) {
    const int row = reg * kRegionRows + it * kClassifyBlock + (int)threadIdx.x;
    const bool is_cell = row < gm.rows_utt && tt[it] < T && uu[it] <= U;
    raw[it].flags = cell_flags(is_cell, tt[it], uu[it], T, U);
    cf[it] = cell_coef<kSparseRows>(raw[it], gm.prune_log2, gm.blank);
    const bool full = cf[it].x > -INFINITY;
    const bool sparse = !full && ((cf[it].y != 0.f) | (cf[it].z != 0.f));
    mask[it] = __ballot(full);
    smask[it] = __ballot(sparse);
    n_cell += __popcll(__ballot(is_cell));
    n_unread += __popcll(__ballot(is_cell && !full));
    if (sparse) {
      const int lab = __float_as_int(cf[it].w);
      if (lab == gm.blank) {
        const float gv = unread_elem(lab, gm.blank, lab, cf[it].y, cf[it].z, sc);
        sq = __builtin_fmaf(gv, gv, sq);
      } else {
        const float gb = unread_elem(gm.blank, gm.blank, lab, cf[it].y, cf[it].z, sc);
        const float gl = unread_elem(lab, gm.blank, lab, cf[it].y, cf[it].z, sc);  // (no label: pl = 0, nothing added)
        sq = __builtin_fmaf(gb, gb, sq);
        sq = __builtin_fmaf(gl, gl, sq);
      }
    }
    if (lane == 0) { wcnt[0][it][wv] = __popcll(mask[it]); wcnt[1][it][wv] = __popcll(smask[it]); }
    // (rowbits holds rows_utt rounded up to a multiple of 64 bits per utterance: a region's last words may lie beyond)
    if ((lane & 31) == 0 && (row >> 5) < w.row_words)
      w.rowbits[(size_t)b * w.row_words + (row >> 5)] = (uint32_t)(lane ? (mask[it] >> 32) : mask[it]);
  }
  sq = group_sum<64>(sq);
  if (lane == 0) { cnt_s[0][wv] = n_cell; cnt_s[1][wv] = n_unread; sq_s[wv] = sq; }
  __syncthreads();
  // Compaction in row order (entries of (iteration, wavefront) pairs before this one, then lanes below this one):
  // the full rows from the front of the region's stretch of crec, the sparse ones from its end backwards.
  // A region's last work item is seldom whole.  Its entries are spread evenly over the wavefronts of k_grad_rmw
  // (wavefront w of that kernel takes entries [w * ng, (w + 1) * ng) of an item: each gets a run of `per_wave`
  // consecutive ones, the rest of its slots stand for nothing), so that the item takes as long as its fullest
  // wavefront needs, not as long as a whole item.
  float4* out = w.crec + (size_t)regid * kRegionStride;
  int total = 0;
#pragma unroll
  for (int it = 0; it < kClassifyIters; ++it)
#pragma unroll
    for (int i = 0; i < NW; ++i) total += wcnt[0][it][i];
  const int n_items = (total + gm.rb - 1) / gm.rb;
  const int last0 = (n_items - 1) * gm.rb;         // first entry of the last item
  const int n_last = total - last0;                // its entries (n_items > 0)
  const int ng = gm.rb / gm.waves;
  const int per_wave = (n_last + gm.waves - 1) / gm.waves;
  int base = 0, sbase = 0;
#pragma unroll
  for (int it = 0; it < kClassifyIters; ++it) {
    int mine = base, smine = sbase;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int c = wcnt[0][it][i], cs = wcnt[1][it][i];
      mine += (i < wv) ? c : 0;
      smine += (i < wv) ? cs : 0;
      base += c;
      sbase += cs;
    }
    const int local_row = it * kClassifyBlock + (int)threadIdx.x;
    const float4 e = make_float4(cf[it].x, cf[it].y, cf[it].z, pack_row_label(local_row, __float_as_int(cf[it].w)));
    const unsigned long long below = (1ull << lane) - 1ull;
    if ((mask[it] >> lane) & 1ull) {
      int pos = mine + __popcll(mask[it] & below);
      if (pos >= last0) {
        const int j = pos - last0;
        pos = last0 + (j / per_wave) * ng + (j % per_wave);
      }
      out[pos] = e;
    }
    if ((smask[it] >> lane) & 1ull) out[kRegionStride - 1 - (smine + __popcll(smask[it] & below))] = e;
  }
  // the last item's slots that no entry went to, and the region's slots of the per-item sums
  if (n_items > 0 && (int)threadIdx.x < gm.rb) {
    const int wslot = (int)threadIdx.x / ng, g = (int)threadIdx.x - wslot * ng;
    if (g >= per_wave || wslot * per_wave + g >= n_last)
      out[last0 + threadIdx.x] = make_float4(-INFINITY, 0.f, 0.f, __int_as_float(-1));
  }
  for (int j = threadIdx.x; j < gm.spb; j += kClassifyBlock) w.partial[(size_t)regid * (gm.spb + 1) + j] = 0.f;
  // One atomic per block hands out the items' places and counts the blocks: the block that finds every other block
  // counted knows the final number of items, stores it for k_grad_rmw and leaves the counter at zero for the next
  // gradient pass on this workspace.  (No fence is involved: the list itself is only read by the next kernel.)
  __shared__ int first_item;
  if (threadIdx.x == 0) {
    const unsigned long long old = atomicAdd(&w.hdr->items_ctr, (1ull << 32) | (unsigned long long)n_items);
    first_item = (int)(old & 0xffffffffull);
    if ((int)(old >> 32) == (int)(gridDim.x * gridDim.y) - 1) {
      w.hdr->n_items_ready = first_item + n_items;
      w.hdr->rmw_next = 0u;
      atomicExch(&w.hdr->items_ctr, 0ull);
    }
    w.rcount[regid] = sbase;  // what k_patch_sparse has to do in this region
    uint2 c = make_uint2(0u, 0u);
    float sqr = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) { c.x += cnt_s[0][i]; c.y += cnt_s[1][i]; sqr += sq_s[i]; }
    w.pcount[regid] = c;
    w.partial[(size_t)regid * (gm.spb + 1) + gm.spb] = sqr;  // the region's last slot: its sparse rows
    if (regid == 0) {
      w.hdr->n_count = (int)(gridDim.x * gridDim.y);
      w.hdr->n_partial = (int)(gridDim.x * gridDim.y) * (gm.spb + 1);
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < n_items) w.items[first_item + threadIdx.x] = (regid << kItemBits) | (int)threadIdx.x;
}

// One wavefront per KiB of an utterance's gradient; row_kib = KiB per row (>= 1).
// grid (ceil(KiB per utterance / kFillWaves), B).
#ifndef RNNT_FILL_WAVES
#define RNNT_FILL_WAVES 4
#endif
constexpr int kFillWaves = RNNT_FILL_WAVES;
__global__ __launch_bounds__(kFillWaves * 64) void k_fill_rows(float* __restrict__ grad, size_t gstride, int kib_utt, int row_kib,
                                                   const uint32_t* __restrict__ rowbits, int row_words, int b0) {
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int kib = (int)blockIdx.x * kFillWaves + wv;  // wave-uniform
  if (kib >= kib_utt) return;
  const int b = b0 + (int)blockIdx.y;
  const int row = row_kib == 1 ? kib : kib / row_kib;
  const uint32_t word = rowbits[(size_t)b * row_words + (row >> 5)];
  if ((word >> (row & 31)) & 1u) return;
  const f4v z = {0.f, 0.f, 0.f, 0.f};
  f4v* dst = reinterpret_cast<f4v*>(grad + (size_t)b * gstride + (size_t)kib * 256) + (threadIdx.x & 63);
  __builtin_nontemporal_store(z, dst);
}

// The blank's and the label's entry of the sparse rows, over k_fill_rows' zeros.  grid (regions per utterance, B).
__global__ __launch_bounds__(256) void k_patch_sparse(float* __restrict__ grad, const float* __restrict__ grad_scale,
                                                      SplitGeom gm, Ws w) {
  const int b = blockIdx.y, reg = blockIdx.x;
  const int regid = b * gm.regions_utt + reg;
  const int n = w.rcount[regid];
  if (n == 0) return;
  const float sc = grad_scale ? grad_scale[b] : 1.f;
  const float4* in = w.crec + (size_t)regid * kRegionStride + (kRegionStride - 1);
  float* greg = grad + (size_t)b * gm.gstride + (size_t)reg * kRegionRows * gm.V;
  for (int j = threadIdx.x; j < n; j += 256) {
    const float4 e = in[-j];
    const int packed = __float_as_int(e.w);
    const int lab = (int)(short)(packed & 0xffff);
    float* gp = greg + (size_t)(packed >> 16) * gm.V;
    gp[gm.blank] = unread_elem(gm.blank, gm.blank, lab, e.y, e.z, sc);
    if (lab >= 0 && lab != gm.blank) gp[lab] = unread_elem(lab, gm.blank, lab, e.y, e.z, sc);
  }
}

// entries per work item and items per region of k_grad_rmw<NC, NG> (one row per wavefront instruction: rows of >= 1 KiB)
template <int NG, int W>
struct RmwShape {
  static constexpr int RB = W * NG;
  static_assert(RB <= kRegionStride - kRegionRows, "a region's entries are padded to a whole work item inside its stretch");
  static constexpr int SPB = (kRegionRows + RB - 1) / RB;
  static_assert(SPB <= kItemMask + 1 && SPB <= kClassifyBlock, "an item's number inside its region is packed into kItemBits bits; a thread of k_classify_rows per item");
};

#ifdef RNNT_RMW_TIMING  // experiment builds: cycles spent per phase, summed over blocks (thread 0's clock)
__device__ unsigned long long g_rmw_dbg[8];
__device__ unsigned long long g_rmw_span[2][1024];  // wall clock (100 MHz) at which a block started / ended
#define RMW_T(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); atomicAdd(&g_rmw_dbg[i], now_ - t_last_); t_last_ = now_; } } while (0)
#else
#define RMW_T(i) do { } while (0)
#endif

template <int NC, int NG, int W>
__global__ __launch_bounds__(W * 64) __attribute__((amdgpu_waves_per_eu(RNNT_RMW_OCC))) void k_grad_rmw(
    const float* x, float* grad, const float* __restrict__ grad_scale, SplitGeom gm, Ws w) {
  constexpr int RB = RmwShape<NG, W>::RB, SPB = RmwShape<NG, W>::SPB;
  static_assert(RB >= kMinRmwRows, "the workspace holds one partial-sum slot per kMinRmwRows entries of a region");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int V = gm.V, blank = gm.blank;
  const int nchunk = V >> 2;
#ifdef RNNT_RMW_TIMING
  unsigned long long t_last_ = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x < 1024) { g_rmw_span[0][blockIdx.x] = wall_clock64(); g_rmw_span[1][blockIdx.x] = wall_clock64(); }
#endif
  const int n_items = w.hdr->n_items_ready;
  if ((int)blockIdx.x >= n_items) return;  // (block-uniform)
  __shared__ float4 coef_buf[2][RB];
  __shared__ float red[W];
  __shared__ int code_s[2];
  // Work items are handed out by a counter as blocks come free.  Blocks do not run at one pace, and fewer of them
  // may be resident than the grid holds (measured: half of a grid of two per compute unit started when the other
  // half had finished, and blocks took 0.4 to 0.9 ms for equal shares): with a fixed share per block the kernel
  // ended when the slowest block had worked its share off.  Thread 0 draws a number two items ahead and looks its
  // code up one item ahead; every thread requests the next item's entries while this item is worked on: they travel
  // during the read phase and are parked in the other half of coef_buf right after it.
  auto code_of = [&](unsigned int idx) { return idx < (unsigned int)n_items ? w.items[idx] : -1; };
  if (threadIdx.x == 0) {
    const unsigned int first = atomicAdd(&w.hdr->rmw_next, 2u);
    code_s[0] = code_of(first);
    code_s[1] = code_of(first + 1u);
  }
  __syncthreads();
  int code = code_s[0];
  int code_next = code_s[1];
  if (code < 0) return;  // (block-uniform) a block that starts when nothing is left
  auto entries_of = [&](int c) {
    float4 e = make_float4(-INFINITY, 0.f, 0.f, __int_as_float(-1));
    if (threadIdx.x < RB) e = w.crec[(size_t)(c >> kItemBits) * kRegionStride + (c & kItemMask) * RB + threadIdx.x];
    return e;
  };
  {
    const float4 first = entries_of(code);
    if (threadIdx.x < RB) coef_buf[0][threadIdx.x] = first;
  }
  __syncthreads();
  RMW_T(0);  // first item's entries have arrived
  for (int cur = 0;; cur ^= 1) {
    const float4* coef_s = coef_buf[cur];
    const int regid = code >> kItemBits, k = code & kItemMask;
    const int b = regid / gm.regions_utt;
    const int reg = regid - b * gm.regions_utt;
    const float sc = grad_scale ? grad_scale[b] : 1.f;
    code = code_next;
    unsigned int ticket = 0u;
    if (threadIdx.x == 0 && code >= 0) ticket = atomicAdd(&w.hdr->rmw_next, 1u);  // (requested with the row loads)
    float4 mine = make_float4(-INFINITY, 0.f, 0.f, __int_as_float(-1));
    if (code >= 0) mine = entries_of(code);
    const size_t reg_off = (size_t)reg * kRegionRows * V;
    const float* xreg = x + (size_t)b * gm.xstride + reg_off;
    float* greg = grad + (size_t)b * gm.gstride + reg_off;

    // ---- read phase: every load of this wave's full rows
    float v[NG][NC][4];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const float4 cf = coef_s[wv * NG + g];
      const bool active = cf.x > -INFINITY;
#ifdef RNNT_RMW_FAKE_CONTIG  // timing experiment only (wrong results): an item's rows as one contiguous run
      const float* rp = xreg + (size_t)((k * RB + wv * NG + g) % kRegionRows) * V;
#else
      const float* rp = xreg + (size_t)(__float_as_int(cf.w) >> 16) * V;  // (not dereferenced for entries that stand for nothing)
#endif
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (active && c < nchunk) {
          load_vec<4>(rp + (size_t)c * 4, v[g][i]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[g][i][q] = 0.f;
        }
      }
    }
    RMW_T(1);  // loads issued
    __syncthreads();  // the block's rows turn from reading to writing together
    RMW_T(2);  // barrier passed (not: loads arrived)
    if (threadIdx.x < RB) coef_buf[cur ^ 1][threadIdx.x] = mine;
    int code_after = -1;  // thread 0: the item after the next one (the number has arrived with the row loads)
    if (threadIdx.x == 0 && code >= 0) code_after = code_of(ticket);
    RMW_T(3);  // next item's entries (and with them, in order, this wave's row loads) have arrived
#if RNNT_RMW_PACE > 0
    // one wait for all row loads ahead of the first store, then a drain after every RNNT_RMW_PACE stores (grad_span.h)
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
      int idx = wv * NG + g;
      asm volatile("" : "+v"(idx));  // (looked up again, not kept across the barrier: see k_grad_span)
      const float4 cf = coef_s[idx];
      const float c2 = cf.x, pb = cf.y, pl = cf.z;
      const int packed = __float_as_int(cf.w);
      const int lab = (int)(short)(packed & 0xffff);
      if (!(c2 > -INFINITY)) continue;  // an entry that stands for nothing
#ifdef RNNT_RMW_FAKE_CONTIG
      float* gp = greg + (size_t)((k * RB + idx) % kRegionRows) * V;
#else
      float* gp = greg + (size_t)(packed >> 16) * V;
#endif
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
          float o[4];
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
          store_vec<4>(gp + (size_t)c * 4, o);
        }
#if RNNT_RMW_PACE > 0
        if ((g * NC + i + 1) % RNNT_RMW_PACE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      }
    }

    RMW_T(4);  // write phase of wave 0
    // the item's sum of squared gradient elements, stored by the item's identity (deterministic)
    sqsum = group_sum<64>(sqsum);
    if (lane == 0) red[wv] = sqsum;
    if (threadIdx.x == 0) code_s[cur] = code_after;
    __syncthreads();  // (also: every wavefront is done with this item's coefficients, and the next item's are in place)
    code_next = code_s[cur];
    if (threadIdx.x == 0) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < W; ++i) s += red[i];
      w.partial[(size_t)regid * (SPB + 1) + k] = s;
    }
    RMW_T(5);  // closing barrier
#ifdef RNNT_RMW_TIMING
    if (threadIdx.x == 0) atomicAdd(&g_rmw_dbg[7], 1ull);
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_rmw_span[1][blockIdx.x] = wall_clock64();
#endif
    if (code < 0) break;  // (block-uniform)
  }
}

}  // namespace
