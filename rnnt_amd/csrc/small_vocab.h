// Part of the single translation unit rnnt_mi355x.hip: the small-vocabulary (V <= 128) variants of the two streaming passes.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "workspace.h"
#include "device_util.h"
#include "cell_coef.h"

namespace {

// ---------------------------------------------------------------------------
// Small-vocabulary variants (V <= 128: the phone-level vocabularies the
// reference trains on, V = 40 / 62, and character sets).  A row is only V*4 <= 512 bytes, so
// splitting it over lanes spends more on cross-lane reductions than on the
// row itself.  Instead a wavefront streams tiles of 64 consecutive rows
// (16*V float4, coalesced) through registers into LDS and then every LANE
// owns one ROW: max, sum of exponentials and the gradient are plain serial
// loops over LDS with no cross-lane traffic at all, and the 64 records of a
// tile leave as one contiguous 1 KiB store.  The next tile's loads are in
// flight while the current one is processed.
// ---------------------------------------------------------------------------
constexpr int kSmallMaxV = 128;
constexpr int kSmallBlock = 128;                // two wavefronts, each with its own LDS tile
constexpr int kSmallWaves = kSmallBlock / 64;
// float4 loads per lane per tile: the kernels are instantiated with NLD = 16 (V <= 64) and 32 (V <= 128)
__host__ __device__ inline int small_nld(int V) { return V <= 64 ? 16 : 32; }

// A tile lives in LDS row by row and every lane walks its own row, so the row stride decides the bank pattern.
// At V = 64 the stride is 256 bytes: all lanes of a 128-bit access on the same four banks (measured 3.2 / 4.4 TB/s
// for the two passes where V = 62 reaches 4.9 / 6.0).  PAD = 1 (V = 64): one float4 of padding per row, 17 float4
// of stride, which spreads 16 lanes over all 64 banks (4.9 / 6.2 TB/s).  Padding the other even-float4 row lengths
// up to 64 measured slower in the gradient pass: the address arithmetic costs 14 registers and with them a resident
// wavefront per SIMD, more than their two-way conflicts do.  Above 64 residency is set by LDS alone (a tile is
// 17-34 KiB), registers are free: PAD = 2 pads every row of an even number of float4 (general row length).
// The two passes keep their own tile layout.  Measured at V = 72 / 80 / 96 / 112 (TB/s, padded against not): first
// pass 4.1 / 3.6 / 3.8 / 4.7 against 4.8 / 5.0 / 4.9 / 4.6, second pass 5.6 / 5.4 / 5.7 / 5.1 against
// 5.6 / 4.1 / 5.4 / 5.0 -- so between 64 and 128 only the second pass pads.
__host__ __device__ inline int small_pad_mode(int V, bool grad_pass) {
  if (V % 4 != 0 || (V / 4) % 2 != 0) return 0;
  if (V == 64) return 1;
  if (V == 128) return 2;
  return (V > 64 && grad_pass) ? 2 : 0;
}
// float4 slots of a wave's LDS tile
__host__ __device__ inline int small_lds_f4(int V, bool grad_pass) {
  return 16 * V + (small_pad_mode(V, grad_pass) ? 64 : 0);
}

struct SmallGeom {
  int B, maxT, maxU1, V, blank;
  int frames;            // frames per utterance in memory = t_off + maxT
  int t_off;             // leading frames of every utterance that take no part in the lattice (a crop along T)
  int ntiles_full;       // tiles of 64 whole rows
  long long total_rows;  // B*frames*maxU1
  float prune_log2;      // cells with log2 gamma below this get a zero gradient row without being re-read
};

// b, T, U are in/out: the lengths are reloaded only when the lane's utterance changes (rarely), so that
// decoding a tile does not wait on memory.
// in_range: the row is a frame of the lattice's [maxT, maxU1] grid (inside the tensor and not a cropped frame);
// t is then its frame index in the crop.
__device__ __forceinline__ void small_decode(long long row, const SmallGeom& g, const int32_t* in_len,
                                             const int32_t* lab_len, int& b, int& t, int& u, int& T, int& U,
                                             bool& in_range) {
  const bool in_tensor = row < g.total_rows;
  const unsigned r = in_tensor ? (unsigned)row : 0u;
  const unsigned per_b = (unsigned)g.frames * (unsigned)g.maxU1;
  const int nb = (int)(r / per_b);
  const unsigned rem = r - (unsigned)nb * per_b;
  const int tf = (int)(rem / (unsigned)g.maxU1);
  u = (int)(rem - (unsigned)tf * (unsigned)g.maxU1);
  t = tf - g.t_off;
  in_range = in_tensor & (t >= 0);
  t = max(t, 0);
  if (nb != b) {
    b = nb;
    utt_dims(in_len, lab_len, b, g.maxT, g.maxU1, T, U);
  }
}

// log-sum-exp2 of one row held in LDS (row-per-lane); VEC = widest aligned access of a row
// Base-2 maximum m2 and log2 of the sum of 2^(x*log2e - m2) of one row; the log-sum-exp is m2 + ls, and a
// log-probability is (x*log2e - m2) - ls, whose error scales with the distance from the maximum.
// kRowChunk floats of a row in LDS, every read issued before the first is used: a lane walks its row alone, so a loop
// of one read and its use per step pays the LDS round trip per step (31 of them per pass at V = 62).  FULL: the chunk
// lies inside the row.  Otherwise the addresses are clamped to the row's last vector and what lies at or beyond V
// comes back as `fill`.
// Measured (V = 62; B=8, T=150, U=40 / B=32, T=500, U=100): first pass 17 -> 9 us / 81 -> 77 us with chunks of 16;
// gradient pass 19 -> 14 us / 136 -> 135 us with chunks of 8 or 4, 15 / 166 us with 16 (its code then spills scalars).
constexpr int kRowChunk = 16;
#ifndef RNNT_SMALL_GRAD_CHUNK
#define RNNT_SMALL_GRAD_CHUNK 8
#endif
constexpr int kGradChunk = RNNT_SMALL_GRAD_CHUNK;
template <int VEC, bool FULL, int CH>
__device__ __forceinline__ void small_row_chunk(const float* rowp, int i, int V, float fill, float (&c)[CH]) {
  static_assert(CH % 4 == 0, "whole vectors of any width");
#pragma unroll
  for (int k = 0; k < CH; k += VEC) {
    const int at = FULL ? i + k : min(i + k, V - VEC);
    if constexpr (VEC == 4) { const float4 q = *reinterpret_cast<const float4*>(rowp + at); c[k] = q.x; c[k + 1] = q.y; c[k + 2] = q.z; c[k + 3] = q.w; }
    else if constexpr (VEC == 2) { const float2 q = *reinterpret_cast<const float2*>(rowp + at); c[k] = q.x; c[k + 1] = q.y; }
    else { c[k] = rowp[at]; }
  }
  if constexpr (!FULL) {
#pragma unroll
    for (int k = 0; k < CH; ++k) c[k] = (i + k - (k % VEC) < V) ? c[k] : fill;  // (uniform; V is a multiple of VEC)
  }
}

template <int VEC>
__device__ __forceinline__ void small_row_lse2(const float* rowp, int V, float& m2, float& ls) {
  float m = -INFINITY;
  float c[kRowChunk];
  int i = 0;
  for (; i + kRowChunk <= V; i += kRowChunk) {
    small_row_chunk<VEC, true, kRowChunk>(rowp, i, V, -INFINITY, c);
#pragma unroll
    for (int k = 0; k < kRowChunk; ++k) m = fmaxf(m, c[k]);
  }
  if (i < V) {
    small_row_chunk<VEC, false, kRowChunk>(rowp, i, V, -INFINITY, c);
#pragma unroll
    for (int k = 0; k < kRowChunk; ++k) m = fmaxf(m, c[k]);
  }
  m2 = m * kLog2e;
  float s = 0.f;
  for (i = 0; i + kRowChunk <= V; i += kRowChunk) {
    small_row_chunk<VEC, true, kRowChunk>(rowp, i, V, -INFINITY, c);
#pragma unroll
    for (int k = 0; k < kRowChunk; ++k) s += fast_exp2(__builtin_fmaf(c[k], kLog2e, -m2));
  }
  if (i < V) {  // (2^-inf = 0 for what lies beyond the row)
    small_row_chunk<VEC, false, kRowChunk>(rowp, i, V, -INFINITY, c);
#pragma unroll
    for (int k = 0; k < kRowChunk; ++k) s += fast_exp2(__builtin_fmaf(c[k], kLog2e, -m2));
  }
  ls = fast_log2(s);
}

// Writes the record of one row (and the guard-frame entries next to frames 0 and maxT-1).
__device__ __forceinline__ void small_write_record(const Ws& w, const SmallGeom& g, int b, int t, int u, float4 rc) {
  float4* frame0 = w.rec + ((size_t)b * (g.maxT + 2) + 1) * g.maxU1;
  frame0[(size_t)t * g.maxU1 + u] = rc;
  const float4 neg = make_float4(kNegF, kNegF, 0.f, 0.f);
  if (t == 0) (frame0 - g.maxU1)[u] = neg;
  if (t == g.maxT - 1) (frame0 + (size_t)g.maxT * g.maxU1)[u] = neg;
}

template <int VEC, int PAD, int NLD>
__global__ __launch_bounds__(kSmallBlock) void k_lse_small(const float* __restrict__ x,
                                                           const int32_t* __restrict__ labels,
                                                           const int32_t* __restrict__ in_len,
                                                           const int32_t* __restrict__ lab_len, SmallGeom g, Ws w) {
  extern __shared__ float4 lds4[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int V = g.V;
  const int tile_f4 = 16 * V;  // 64 rows * V floats / 4
  float4* tile4 = lds4 + wv * (tile_f4 + (PAD ? 64 : 0));
  const float* rowp = reinterpret_cast<const float*>(tile4) + lane * (V + (PAD ? 4 : 0));
  // float4 number q of the tile in memory order -> its LDS slot: q + (row of q) when rows are padded.  PAD = 1:
  // 16 float4 per row; PAD = 2: row = q / (V/4) as a multiply by the rounded-up reciprocal (exact for q < 2048,
  // V/4 <= 32)
  const unsigned recip = PAD == 2 ? (65536u + (unsigned)(V / 4) - 1u) / (unsigned)(V / 4) : 0u;
  auto slot = [&](int q) {
    return PAD == 1 ? q + (q >> 4) : PAD == 2 ? q + (int)(__umul24((unsigned)q, recip) >> 16) : q;
  };
  const int gw = blockIdx.x * kSmallWaves + wv, nw = gridDim.x * kSmallWaves;
  const f4v* x4 = reinterpret_cast<const f4v*>(x);

  int T = 0;
  auto row_valid = [&](int n, int& b, int& t, int& u, int& U, bool& in_range) {
    small_decode((long long)n * 64 + lane, g, in_len, lab_len, b, t, u, T, U, in_range);
    return in_range & (t < T) & (u <= U);
  };

  // Every pass issues the same NLD loads: a tile without a lattice cell is not skipped but replaced by a re-read of
  // the last tile this wave did load (its own, recent addresses: served by the caches, and no address that every
  // wave would hit at once), and the lanes past the end of a tile re-read its last 16 bytes.  With conditional loads
  // the compiler cannot count what is outstanding.
  f4v st[NLD];
  int n_read = gw < g.ntiles_full ? gw : 0;
  auto issue = [&](int n, bool any) {
    n_read = any ? n : n_read;
    const f4v* src = x4 + (size_t)n_read * tile_f4;
#pragma unroll
    for (int i = 0; i < NLD; ++i) st[i] = __builtin_nontemporal_load(src + min(lane + 64 * i, tile_f4 - 1));
  };

  // The counter that memory operations are waited on is in-order: a wait for one load also waits for everything
  // issued before it, and a wait the compiler cannot count (conditional stores in flight) waits for everything.
  // So nothing in the loop waits except its top, and what is outstanding there is what it needs: a tile's label is
  // loaded ahead of its logits, at decode time, and a tile's record is stored one pass later, after the next tile
  // has been staged and before the loads of the one after go out (a label looked up after the row's sum, and the
  // records stored at the end of the pass, made every pass wait twice for loads it had only just issued).
  int n = gw;
  int b = -1, t = 0, u = 0, U = 0, lab = 0;
  bool in_range = false;
  bool valid = false, any = false;
  auto label_of = [&](bool ok, int bb, int uu, int UU) {
    return (ok && uu < UU) ? labels[(size_t)bb * (g.maxU1 - 1) + uu] : 0;
  };
  if (n < g.ntiles_full) {
    valid = row_valid(n, b, t, u, U, in_range);
    any = __ballot(valid) != 0ull;
    lab = label_of(valid, b, u, U);
    issue(n, any);
  }
  int pb = 0, pt = 0, pu = 0;  // the previous tile's row and record, not stored yet
  bool pin = false;
  float4 prc = make_float4(kNegF, kNegF, 0.f, 0.f);
  for (; n < g.ntiles_full; n += nw) {
    // stage the tile into LDS (only this wavefront touches it: in-order LDS, no barrier needed)
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int q = min(lane + 64 * i, tile_f4 - 1);
      tile4[slot(q)] = make_float4(st[i].x, st[i].y, st[i].z, st[i].w);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (pin) small_write_record(w, g, pb, pt, pu, prc);
    __builtin_amdgcn_sched_barrier(0);
    const int cb = b, ct = t, cu = u, cU = U;
    int clab = lab;
    const bool cvalid = valid, cany = any, cin = in_range;
    // next tile: decode and put its loads in flight before working on this one
    const int nn = n + nw;
    if (nn < g.ntiles_full) {
      valid = row_valid(nn, b, t, u, U, in_range);
      any = __ballot(valid) != 0ull;
      lab = label_of(valid, b, u, U);
      issue(nn, any);
    }
    __builtin_amdgcn_sched_barrier(0);
    float4 rc = make_float4(kNegF, kNegF, 0.f, 0.f);
    if (cany && cvalid) {
      float m2, ls;
      small_row_lse2<VEC>(rowp, V, m2, ls);
      float lpl = kNegF;
      if (cu < cU) {
        if (clab < 0 || clab >= V) {
          atomicOr(&w.hdr->flags, RNNT_MI355X_FLAG_BAD_LABEL);
          clab = 0;
        }
        lpl = __builtin_fmaf(rowp[clab], kLog2e, -m2) - ls;
      }
      rc = make_float4(__builtin_fmaf(rowp[g.blank], kLog2e, -m2) - ls, lpl, m2 + ls, -ls);
    }
    pb = cb; pt = ct; pu = cu; pin = cin; prc = rc;
  }
  if (pin) small_write_record(w, g, pb, pt, pu, prc);
  // the last, partial tile (fewer than 64 rows): one wavefront reads its rows straight from global memory
  if (gw == 0 && (long long)g.ntiles_full * 64 < g.total_rows) {
    const long long row = (long long)g.ntiles_full * 64 + lane;
    b = -1;
    small_decode(row, g, in_len, lab_len, b, t, u, T, U, in_range);
    if (in_range) {
      float4 rc = make_float4(kNegF, kNegF, 0.f, 0.f);
      if (t < T && u <= U) {
        const float* rp = x + (size_t)row * V;
        float m2, ls;
        small_row_lse2<1>(rp, V, m2, ls);
        float lpl = kNegF;
        if (u < U) {
          int lab = labels[(size_t)b * (g.maxU1 - 1) + u];
          if (lab < 0 || lab >= V) {
            atomicOr(&w.hdr->flags, RNNT_MI355X_FLAG_BAD_LABEL);
            lab = 0;
          }
          lpl = __builtin_fmaf(rp[lab], kLog2e, -m2) - ls;
        }
        rc = make_float4(__builtin_fmaf(rp[g.blank], kLog2e, -m2) - ls, lpl, m2 + ls, -ls);
      }
      small_write_record(w, g, b, t, u, rc);
    }
  }
}

// ---------------------------------------------------------------------------
// k_lse_small_span: k_lse_small with one tile per wavefront and no loop (see k_grad_small_span below: a block is one
// wavefront that handles one tile and exits, so nothing it stores is ever waited for, and the hardware's block
// scheduler does the pipelining).  Lengths and label first (a short round trip: a few addresses per tile, mostly
// cache hits), then the tile's rows if one of them holds a cell.
// ---------------------------------------------------------------------------
template <int VEC, int PAD, int NLD>
__global__ __launch_bounds__(64) void k_lse_small_span(const float* __restrict__ x, const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ in_len,
                                                       const int32_t* __restrict__ lab_len, SmallGeom g, Ws w) {
  extern __shared__ float4 lds4[];
  const int lane = threadIdx.x;
  const int V = g.V;
  const int tile_f4 = 16 * V;
  float4* tile4 = lds4;
  const float* rowp = reinterpret_cast<const float*>(tile4) + lane * (V + (PAD ? 4 : 0));
  const unsigned recip = PAD == 2 ? (65536u + (unsigned)(V / 4) - 1u) / (unsigned)(V / 4) : 0u;
  auto slot = [&](int q) {  // see k_lse_small
    return PAD == 1 ? q + (q >> 4) : PAD == 2 ? q + (int)(__umul24((unsigned)q, recip) >> 16) : q;
  };
  const f4v* x4 = reinterpret_cast<const f4v*>(x);
  const bool has_tail = (long long)g.ntiles_full * 64 < g.total_rows;
  const int n = (int)blockIdx.x - (has_tail ? 1 : 0);
  if (n < 0) {  // the last, partial tile (fewer than 64 rows): rows straight from global memory
    const long long row = (long long)g.ntiles_full * 64 + lane;
    int b = -1, t, u, T = 0, U = 0;
    bool in_range;
    small_decode(row, g, in_len, lab_len, b, t, u, T, U, in_range);
    if (in_range) {
      float4 rc = make_float4(kNegF, kNegF, 0.f, 0.f);
      if (t < T && u <= U) {
        const float* rp = x + (size_t)row * V;
        float m2, ls;
        small_row_lse2<1>(rp, V, m2, ls);
        float lpl = kNegF;
        if (u < U) {
          int lab = labels[(size_t)b * (g.maxU1 - 1) + u];
          if (lab < 0 || lab >= V) {
            atomicOr(&w.hdr->flags, RNNT_MI355X_FLAG_BAD_LABEL);
            lab = 0;
          }
          lpl = __builtin_fmaf(rp[lab], kLog2e, -m2) - ls;
        }
        rc = make_float4(__builtin_fmaf(rp[g.blank], kLog2e, -m2) - ls, lpl, m2 + ls, -ls);
      }
      small_write_record(w, g, b, t, u, rc);
    }
    return;
  }
  const unsigned rr = (unsigned)n * 64u + (unsigned)lane;  // < total_rows < 2^31
  const unsigned per_b = (unsigned)g.frames * (unsigned)g.maxU1;
  const int b = (int)(rr / per_b);
  const unsigned rem = rr - (unsigned)b * per_b;
  const int tf = (int)(rem / (unsigned)g.maxU1);
  const int u = (int)(rem - (unsigned)tf * (unsigned)g.maxU1);
  const bool in_range = tf >= g.t_off;
  const int t = max(tf - g.t_off, 0);
  const int Traw = in_len[b], Uraw = lab_len[b];
  int lab = 0;
  if (g.maxU1 > 1) lab = labels[(size_t)b * (g.maxU1 - 1) + min(u, g.maxU1 - 2)];
  const bool ok = (Traw >= 1) & (Traw <= g.maxT) & (Uraw >= 0) & (Uraw <= g.maxU1 - 1);
  const int T = ok ? Traw : 0, U = ok ? Uraw : 0;
  const bool valid = in_range & (t < T) & (u <= U);
  float4 rc = make_float4(kNegF, kNegF, 0.f, 0.f);
  if (__ballot(valid) != 0ull) {  // wave-uniform
    const f4v* src = x4 + (size_t)n * tile_f4;
    f4v st[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) st[i] = __builtin_nontemporal_load(src + min(lane + 64 * i, tile_f4 - 1));
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int q = min(lane + 64 * i, tile_f4 - 1);
      tile4[slot(q)] = make_float4(st[i].x, st[i].y, st[i].z, st[i].w);
    }
    // (one wavefront: its LDS accesses are in program order, no barrier)
    if (valid) {
      float m2, ls;
      small_row_lse2<VEC>(rowp, V, m2, ls);
      float lpl = kNegF;
      if (u < U) {
        if (lab < 0 || lab >= V) {
          atomicOr(&w.hdr->flags, RNNT_MI355X_FLAG_BAD_LABEL);
          lab = 0;
        }
        lpl = __builtin_fmaf(rowp[lab], kLog2e, -m2) - ls;
      }
      rc = make_float4(__builtin_fmaf(rowp[g.blank], kLog2e, -m2) - ls, lpl, m2 + ls, -ls);
    }
  }
  if (in_range) small_write_record(w, g, b, t, u, rc);
}

// The cell state of one row (CellRaw, cell_coef.h), loaded only where the row holds a lattice cell.
__device__ __forceinline__ CellRaw cell_load(const Ws& w, const int32_t* labels, int b, int t, int u, int T, int U,
                                             int maxT, int maxU1, int V, bool in_range) {
  CellRaw r;
  r.a = r.be = r.be_t = r.be_u = r.ll = 0.0;
  r.rc = make_float4(0.f, 0.f, 0.f, 0.f);
  r.lab = -1;
  r.flags = 0;
  if (in_range && t < T && u <= U) {
    const int U1s = w.U1s;
    const size_t ci = ((size_t)b * w.D + (t + u)) * U1s + u;
    r.flags = cell_flags(true, t, u, T, U);
    r.a = w.alpha[ci];
    r.be = w.beta[ci];
    r.ll = w.ll[b];
    r.rc = w.rec[((size_t)b * (maxT + 2) + 1 + t) * maxU1 + u];
    if (t < T - 1) r.be_t = w.beta[ci + U1s];
    if (u < U) {
      r.be_u = w.beta[ci + U1s + 1];
      const int l = labels[(size_t)b * (maxU1 - 1) + u];
      r.lab = (l < 0 || l >= V) ? 0 : l;
    }
  }
  return r;
}

template <int VEC>
__device__ __forceinline__ float small_row_grad(const float* src, float* dst, int V, int blank, float4 cf, float sc) {
  const float c2 = cf.x, pb = cf.y, pl = cf.z;
  const int lab = __float_as_int(cf.w);
  float sq = 0.f;
  float c[kGradChunk];
  auto work = [&](int i) {  // (an element beyond the row arrives as -inf and leaves as 0: it is neither blank nor label)
#pragma unroll
    for (int k = 0; k < kGradChunk; ++k) {
      float gv = fast_exp2(__builtin_fmaf(c[k], kLog2e, c2));
      gv -= (i + k == blank) ? pb : 0.f;
      gv -= (i + k == lab) ? pl : 0.f;
      gv *= sc;
      sq = __builtin_fmaf(gv, gv, sq);
      c[k] = gv;
    }
  };
  auto put = [&](int at, int k) {
    if constexpr (VEC == 4) *reinterpret_cast<float4*>(dst + at) = make_float4(c[k], c[k + 1], c[k + 2], c[k + 3]);
    else if constexpr (VEC == 2) *reinterpret_cast<float2*>(dst + at) = make_float2(c[k], c[k + 1]);
    else dst[at] = c[k];
  };
  int i = 0;
  for (; i + kGradChunk <= V; i += kGradChunk) {
    small_row_chunk<VEC, true, kGradChunk>(src, i, V, -INFINITY, c);
    work(i);
#pragma unroll
    for (int k = 0; k < kGradChunk; k += VEC) put(i + k, k);
  }
  if (i < V) {
    small_row_chunk<VEC, false, kGradChunk>(src, i, V, -INFINITY, c);
    work(i);
#pragma unroll
    for (int k = 0; k < kGradChunk; k += VEC)
      if (i + k < V) put(i + k, k);  // (uniform)
  }
  return sq;
}

template <int VEC, int PAD, int NLD>
__global__ __launch_bounds__(kSmallBlock) void k_grad_small(const float* x, float* grad,
                                                            const int32_t* __restrict__ labels,
                                                            const int32_t* __restrict__ in_len,
                                                            const int32_t* __restrict__ lab_len,
                                                            const float* __restrict__ grad_scale, SmallGeom g, Ws w) {
  extern __shared__ float4 lds4[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int V = g.V;
  const int tile_f4 = 16 * V;
  float4* tile4 = lds4 + wv * (tile_f4 + (PAD ? 64 : 0));
  float* rowp = reinterpret_cast<float*>(tile4) + lane * (V + (PAD ? 4 : 0));
  const unsigned recip = PAD == 2 ? (65536u + (unsigned)(V / 4) - 1u) / (unsigned)(V / 4) : 0u;
  auto slot = [&](int q) {  // see k_lse_small
    return PAD == 1 ? q + (q >> 4) : PAD == 2 ? q + (int)(__umul24((unsigned)q, recip) >> 16) : q;
  };
  const int gw = blockIdx.x * kSmallWaves + wv, nw = gridDim.x * kSmallWaves;
  const f4v* x4 = reinterpret_cast<const f4v*>(x);
  f4v* g4 = reinterpret_cast<f4v*>(grad);
  if (blockIdx.x == 0 && threadIdx.x == 0) { w.hdr->n_partial = gridDim.x; w.hdr->n_count = gridDim.x; }
  __shared__ int tail_cnt[2];  // cells / pruned cells of the last, partial tile (block 0 only)
  if (threadIdx.x < 2) tail_cnt[threadIdx.x] = 0;
  __syncthreads();

  // rows [row_lo(i), row_hi(i)] of a tile overlap its i-th KiB: that load is needed iff one of them is active
  auto chunk_needed = [&](unsigned long long active, int i) {
    const int row_bytes = 4 * V;
    const int lo = (1024 * i) / row_bytes;
    const int hi = min(63, (1024 * i + 1023) / row_bytes);
    const unsigned long long m = (hi >= 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~((1ull << lo) - 1ull);
    return (active & m) != 0ull;
  };

  int cb = -1, cT = 0, cU = 0;  // lengths cache of small_decode
  auto raw_of = [&](int n, float& sc) {
    int t, u;
    bool in_range;
    small_decode((long long)n * 64 + lane, g, in_len, lab_len, cb, t, u, cT, cU, in_range);
    sc = grad_scale ? grad_scale[cb] : 1.f;
    return cell_load(w, labels, cb, t, u, cT, cU, g.maxT, g.maxU1, V, in_range);
  };

  f4v st[NLD];
  auto issue = [&](int n, unsigned long long active) {
    const f4v* src = x4 + (size_t)n * tile_f4;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int q = lane + 64 * i;
      if (chunk_needed(active, i) && q < tile_f4) st[i] = __builtin_nontemporal_load(src + q);
    }
  };

  float sqsum = 0.f;
  int n_cell = 0, n_pruned = 0;  // wave-uniform counts of lattice cells seen / pruned
  auto count = [&](const CellRaw& r, const float4& c) {
    n_cell += __popcll(__ballot(r.flags & kCellValid));
    n_pruned += __popcll(__ballot((r.flags & kCellValid) && !(c.x > -INFINITY)));
  };
  // Tiles are taken in the reverse of k_lse_small's order (most recently read rows first).  Three stages are in
  // flight: the cell state of tile k+2, the logits of tile k+1 (gated by which of its rows are active) and the
  // arithmetic of tile k.
  const int last = g.ntiles_full - 1;
  int k = gw;
  float4 cf = make_float4(-INFINITY, 0.f, 0.f, __int_as_float(-1));
  float sc = 1.f, sc_raw = 1.f;
  unsigned long long active = 0ull;
  CellRaw raw;
  raw.flags = 0;
  if (k < g.ntiles_full) {
    const CellRaw r0 = raw_of(last - k, sc);
    cf = cell_coef<false>(r0, g.prune_log2, g.blank);
    count(r0, cf);
    active = __ballot(cf.x > -INFINITY);
    issue(last - k, active);
    if (k + nw < g.ntiles_full) raw = raw_of(last - (k + nw), sc_raw);
  }
  for (; k < g.ntiles_full; k += nw) {
    const int n = last - k;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int q = lane + 64 * i;
      if (chunk_needed(active, i) && q < tile_f4) tile4[slot(q)] = make_float4(st[i].x, st[i].y, st[i].z, st[i].w);
    }
    const float4 ccf = cf;
    const float csc = sc;
    if (k + nw < g.ntiles_full) {
      cf = cell_coef<false>(raw, g.prune_log2, g.blank);  // loaded one iteration ago
      count(raw, cf);
      sc = sc_raw;
      active = __ballot(cf.x > -INFINITY);
      issue(last - (k + nw), active);
      if (k + 2 * nw < g.ntiles_full) raw = raw_of(last - (k + 2 * nw), sc_raw);
    }
    if (ccf.x > -INFINITY) {
      sqsum += small_row_grad<VEC>(rowp, rowp, V, g.blank, ccf, csc);
    } else {
      for (int i = 0; i < V; i += VEC) {
        if constexpr (VEC == 4) *reinterpret_cast<float4*>(rowp + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        else if constexpr (VEC == 2) *reinterpret_cast<float2*>(rowp + i) = make_float2(0.f, 0.f);
        else rowp[i] = 0.f;
      }
    }
    // the whole tile (gradients and zero rows) leaves as coalesced 1 KiB stores
    f4v* dst = g4 + (size_t)n * tile_f4;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int q = lane + 64 * i;
      if (q < tile_f4) {
        const float4 o = tile4[slot(q)];
        f4v ov = {o.x, o.y, o.z, o.w};
        __builtin_nontemporal_store(ov, dst + q);
      }
    }
  }
  // the last, partial tile: rows handled one per lane straight in global memory
  if (gw == 0 && (long long)g.ntiles_full * 64 < g.total_rows) {
    const long long row = (long long)g.ntiles_full * 64 + lane;
    int b = -1, t, u, T = 0, U = 0;
    bool in_range;
    small_decode(row, g, in_len, lab_len, b, t, u, T, U, in_range);
    if (row < g.total_rows) {
      const CellRaw rl = cell_load(w, labels, b, t, u, T, U, g.maxT, g.maxU1, V, in_range);
      const float4 c = cell_coef<false>(rl, g.prune_log2, g.blank);
      if (rl.flags & kCellValid) {  // divergent here: counted per lane, summed below
        atomicAdd(&tail_cnt[0], 1);
        if (!(c.x > -INFINITY)) atomicAdd(&tail_cnt[1], 1);
      }
      float* gp = grad + (size_t)row * V;
      if (c.x > -INFINITY) {
        sqsum += small_row_grad<1>(x + (size_t)row * V, gp, V, g.blank, c, grad_scale ? grad_scale[b] : 1.f);
      } else {
        for (int i = 0; i < V; ++i) gp[i] = 0.f;
      }
    }
  }
  __shared__ float red[kSmallWaves];
  __shared__ int cnt[2][kSmallWaves];
  sqsum = group_sum<64>(sqsum);
  if (lane == 0) { red[wv] = sqsum; cnt[0][wv] = n_cell; cnt[1][wv] = n_pruned; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    uint2 c = make_uint2((unsigned)tail_cnt[0], (unsigned)tail_cnt[1]);
#pragma unroll
    for (int i = 0; i < kSmallWaves; ++i) { s += red[i]; c.x += cnt[0][i]; c.y += cnt[1][i]; }
    w.partial[blockIdx.x] = s;
    w.pcount[blockIdx.x] = c;
  }
}

// ---------------------------------------------------------------------------
// k_grad_small_span: the arithmetic and the tile layout of k_grad_small, one tile per wavefront and no loop.
//
// k_grad_small's persistent waves go from tile to tile, and because its loads and stores are conditional the
// compiler cannot count them: every wait in its loop is s_waitcnt vmcnt(0), i.e. a wave drains the previous tile's
// stores before it touches the next tile (about 14 us per tile and wave under load, with 10 waves per compute unit
// -- as many as their 16 KiB LDS tiles allow).  A wavefront that ends may leave its stores in flight.  So here a
// block is one wavefront that handles exactly one tile and exits; the lengths and the whole cell state are requested
// in one round trip (clamped addresses), the gated rows in a second.  Measured at B=32, T=500, U=100, V=62: 141 ->
// 130 us on N(0,1) logits, 132 -> 119 us on peaked ones, 176 -> 151 us without pruning.  (Blocks of 2 or 4 such
// wavefronts with a barrier between their loads and their stores: the same or slower.  Wavefronts of a block taking
// turns on one LDS tile, so that residency is set by registers instead of LDS, 16-20 waves per compute unit: slower.)
// ---------------------------------------------------------------------------
template <int VEC, int PAD, int NLD>
__global__ __launch_bounds__(64) void k_grad_small_span(const float* x, float* grad, const int32_t* __restrict__ labels,
                                                        const int32_t* __restrict__ in_len,
                                                        const int32_t* __restrict__ lab_len,
                                                        const float* __restrict__ grad_scale, SmallGeom g, Ws w) {
  extern __shared__ float4 lds4[];
  const int lane = threadIdx.x;
  const int V = g.V;
  const int tile_f4 = 16 * V;
  float4* tile4 = lds4;
  float* rowp = reinterpret_cast<float*>(tile4) + lane * (V + (PAD ? 4 : 0));
  const unsigned recip = PAD == 2 ? (65536u + (unsigned)(V / 4) - 1u) / (unsigned)(V / 4) : 0u;
  auto slot = [&](int q) {  // see k_lse_small
    return PAD == 1 ? q + (q >> 4) : PAD == 2 ? q + (int)(__umul24((unsigned)q, recip) >> 16) : q;
  };
  const f4v* x4 = reinterpret_cast<const f4v*>(x);
  f4v* g4 = reinterpret_cast<f4v*>(grad);
  if (blockIdx.x == 0 && lane == 0) { w.hdr->n_partial = gridDim.x; w.hdr->n_count = gridDim.x; }

  // rows [row_lo(i), row_hi(i)] of a tile overlap its i-th KiB: that load is needed iff one of them is active
  auto chunk_needed = [&](unsigned long long active, int i) {
    const int row_bytes = 4 * V;
    const int lo = (1024 * i) / row_bytes;
    const int hi = min(63, (1024 * i + 1023) / row_bytes);
    const unsigned long long m = (hi >= 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~((1ull << lo) - 1ull);
    return (active & m) != 0ull;
  };

  // tiles in the reverse of k_lse_small's order (most recently read rows first).  The grid has one block per whole
  // tile, and one more when the tensor ends in a partial tile (block 0 then takes those rows, one per lane, straight
  // in global memory).
  const bool has_tail = (long long)g.ntiles_full * 64 < g.total_rows;
  const int k = (int)blockIdx.x - (has_tail ? 1 : 0);
  float sqsum = 0.f;
  int n_cell = 0, n_pruned = 0;
  if (k < 0) {
    const long long row = (long long)g.ntiles_full * 64 + lane;
    int b = -1, t, u, T = 0, U = 0;
    bool in_range;
    small_decode(row, g, in_len, lab_len, b, t, u, T, U, in_range);
    bool cell = false, pruned = false;
    if (row < g.total_rows) {
      const CellRaw rl = cell_load(w, labels, b, t, u, T, U, g.maxT, g.maxU1, V, in_range);
      const float4 c = cell_coef<false>(rl, g.prune_log2, g.blank);
      cell = rl.flags & kCellValid;
      pruned = cell && !(c.x > -INFINITY);
      float* gp = grad + (size_t)row * V;
      if (c.x > -INFINITY) {
        sqsum += small_row_grad<1>(x + (size_t)row * V, gp, V, g.blank, c, grad_scale ? grad_scale[b] : 1.f);
      } else {
        for (int i = 0; i < V; ++i) gp[i] = 0.f;
      }
    }
    n_cell = __popcll(__ballot(cell));
    n_pruned = __popcll(__ballot(pruned));
  } else {
    const int n = g.ntiles_full - 1 - k;
    // The lengths and everything a cell needs are requested at once (addresses clamped into the arrays; what is
    // loaded for a row without a cell is not used): one round trip to memory ahead of the row loads, not two.
    const unsigned rr = (unsigned)n * 64u + (unsigned)lane;  // < total_rows < 2^31
    const unsigned per_b = (unsigned)g.frames * (unsigned)g.maxU1;
    const int b = (int)(rr / per_b);
    const unsigned rem = rr - (unsigned)b * per_b;
    const int tf = (int)(rem / (unsigned)g.maxU1);
    const int u = (int)(rem - (unsigned)tf * (unsigned)g.maxU1);
    const bool in_range = tf >= g.t_off;
    const int t = max(tf - g.t_off, 0);
    const int U1s = w.U1s;
    const size_t ci = ((size_t)b * w.D + (t + u)) * U1s + u;
    const size_t ci_t = (t + u + 1 < w.D) ? ci + U1s : ci;
    const size_t ci_u = (u + 1 < U1s) ? ci_t + 1 : ci_t;
    const int Traw = in_len[b], Uraw = lab_len[b];
    CellRaw r0;
    r0.a = w.alpha[ci];
    r0.be = w.beta[ci];
    r0.be_t = w.beta[ci_t];
    r0.be_u = w.beta[ci_u];
    r0.ll = w.ll[b];
    r0.rc = w.rec[((size_t)b * (g.maxT + 2) + 1 + t) * g.maxU1 + u];
    int lab = 0;
    if (g.maxU1 > 1) lab = labels[(size_t)b * (g.maxU1 - 1) + min(u, g.maxU1 - 2)];
    const float sc = grad_scale ? grad_scale[b] : 1.f;
    const bool ok = (Traw >= 1) & (Traw <= g.maxT) & (Uraw >= 0) & (Uraw <= g.maxU1 - 1);
    const int T = ok ? Traw : 0, U = ok ? Uraw : 0;
    const bool is_cell = in_range && t < T && u <= U;
    r0.flags = cell_flags(is_cell, t, u, T, U);
    r0.lab = (u < U) ? ((lab < 0 || lab >= V) ? 0 : lab) : -1;
    const float4 cf = cell_coef<false>(r0, g.prune_log2, g.blank);
    n_cell = __popcll(__ballot(r0.flags & kCellValid));
    n_pruned = __popcll(__ballot((r0.flags & kCellValid) && !(cf.x > -INFINITY)));
    const unsigned long long active = __ballot(cf.x > -INFINITY);
    f4v* dst = g4 + (size_t)n * tile_f4;
    if (active == 0ull) {  // no active row: zeros straight to memory
      const f4v z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int q = lane + 64 * i;
        if (q < tile_f4) __builtin_nontemporal_store(z, dst + q);
      }
    } else {
      const f4v* src = x4 + (size_t)n * tile_f4;
      f4v st[NLD];
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int q = lane + 64 * i;
        if (chunk_needed(active, i) && q < tile_f4) st[i] = __builtin_nontemporal_load(src + q);
      }
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int q = lane + 64 * i;
        if (chunk_needed(active, i) && q < tile_f4) tile4[slot(q)] = make_float4(st[i].x, st[i].y, st[i].z, st[i].w);
      }
      // (one wavefront: its LDS accesses are in program order, no barrier)
      if (cf.x > -INFINITY) {
        sqsum += small_row_grad<VEC>(rowp, rowp, V, g.blank, cf, sc);
      } else {
        for (int i = 0; i < V; i += VEC) {
          if constexpr (VEC == 4) *reinterpret_cast<float4*>(rowp + i) = make_float4(0.f, 0.f, 0.f, 0.f);
          else if constexpr (VEC == 2) *reinterpret_cast<float2*>(rowp + i) = make_float2(0.f, 0.f);
          else rowp[i] = 0.f;
        }
      }
      // the whole tile (gradients and zero rows) leaves as coalesced 1 KiB stores
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int q = lane + 64 * i;
        if (q < tile_f4) {
          const float4 o = tile4[slot(q)];
          f4v ov = {o.x, o.y, o.z, o.w};
          __builtin_nontemporal_store(ov, dst + q);
        }
      }
    }
  }
  sqsum = group_sum<64>(sqsum);
  if (lane == 0) {
    w.partial[blockIdx.x] = sqsum;
    w.pcount[blockIdx.x] = make_uint2((unsigned)n_cell, (unsigned)n_pruned);
  }
}

}  // namespace
