// Part of the single translation unit rnnt_mi355x.hip: the additive joint fused into the loss (matrix-core contractions).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "workspace.h"
#include "device_util.h"
#include "small_vocab.h"

namespace {

// ---------------------------------------------------------------------------
// Fused additive joint (reference Util.rnnt_lambda_func, rnnt/base/util.py:
// 107-109: acts = log_softmax(y_trans[B,T,1,V] + y_pred[B,1,U+1,V])).  The
// [B,T,U+1,V] tensor is never formed.  With f = y_trans, g = y_pred and
//   F(t,v) = exp(f(t,v) - max_v f(t,.)),   G(u,v) = exp(g(u,v) - max_v g(u,.))
// the sum over the vocabulary of every cell is a matrix product,
//   S(t,u) = sum_v F(t,v) G(u,v),      lse(t,u) = mf(t) + mg(u) + log S(t,u),
// and so are the two gradients: with C(t,u) = s_b * gamma(t,u) / S(t,u)
//   d f(t,v) = F(t,v) * sum_u C(t,u) G(u,v)  - (blank / label terms)
//   d g(u,v) = G(u,v) * sum_t C(t,u) F(t,v)  - (blank / label terms)
// Three batched float32 contractions on the matrix cores
// (v_mfma_f32_32x32x2_f32, operands straight from L2-resident global memory)
// replace T*U*V exponentials.  Cells whose S underflows (maxima of f and g at
// different v, far apart) are recomputed with the exact maximum, flagged, and
// their gradient goes through the exponent-domain kernels further down.
//   k_joint_prep    mf2, mg2 (base-2 row maxima), F, G, the blank / label entries of every row
//   k_joint_gather  f2(t, y_{u+1}) of every cell; final "log 0" records of everything that is no lattice cell
//   k_joint_lse_mm  S -> records {lpb, lpl, log2 S}          (then k_sweep as usual)
//   k_joint_coef    {c, pb, pl, label code} and C of every cell from alpha, beta, ll
//   k_joint_df_mm, k_joint_dg_mm   the two gradient contractions
//   k_joint_labelmap               chains of label positions per symbol: the blank / label terms of d f are
//                                  subtracted in the epilogue of its contraction
//   k_joint_corr_g                 the blank / label terms of d g
//   k_joint_fix                    exponent-domain contributions of flagged cells to d f and d g (exits at once
//                                  when there are none)
// ---------------------------------------------------------------------------
struct JointGeom {
  int B, maxT, maxU1, V, blank;
  size_t fstride;  // elements between utterances of f (>= maxT*V): lets f be a crop along T of a larger tensor
  size_t dfstride; // same for d f
  float prune_log2 = kPruneLog2;  // cells whose occupancy is below 2^prune_log2 are left out of the gradient (backward only)
};

typedef float f16v __attribute__((ext_vector_type(16)));
constexpr float kJointMinSum = 5.421010862427522e-20f;  // 2^-64: below this the linear-domain sum is not trusted
constexpr int kMmSets = 2;  // operand register sets per wavefront in the contraction loops
constexpr int kJointFlagBit = 1 << 20;                  // label code = (label + 1) | flag; labels < 4096

__device__ __forceinline__ f4v ld4(const float* p) { return *reinterpret_cast<const f4v*>(p); }

// Offset of element (row, col) of a matrix stored in tiles of 32 rows x 8 columns, `chunks` = padded columns / 8.
// A wavefront whose lanes take rows r0..r0+31 and 4 consecutive columns each (lanes 0-31 columns c..c+3, lanes
// 32-63 columns c+4..c+7) reads 1 KiB of consecutive addresses; 8 consecutive rows of 8 columns are 2 lines.
__device__ __forceinline__ size_t tiled_off(int row, int col, int chunks) {
  return ((((size_t)(row >> 5) * chunks) + (col >> 3)) << 8) + ((row & 31) << 3) + (col & 7);
}

// out[k] = row[lane+64k]*log2(e) - m2, or -inf past the end of the row
template <int KE>
__device__ __forceinline__ void jrow_load(const float* row, int V, int lane, float m2, float (&out)[KE]) {
#pragma unroll
  for (int k = 0; k < KE; ++k) {
    const int v = lane + 64 * k;
    const float x = row[min(v, V - 1)];  // unconditional load (a guarded one costs a branch and a wait per element)
    out[k] = v < V ? __builtin_fmaf(x, kLog2e, -m2) : -INFINITY;
  }
}

// One wavefront per 4 consecutive (padded) rows of f or g: their base-2 maxima, then their part of F or G in the
// tiled layout.  Lane = (row q of 4, column chunk cc of 8, half e4 of a chunk): one step reads 256 contiguous
// bytes of each of the 4 rows and writes 8 whole 128-byte lines.  Padding rows and columns become zeros.
// VEC4: rows start on 16-byte boundaries (V % 4 == 0, aligned bases), so the reads are float4.
// NIT > 0: the rows stay in registers between the two passes (NIT float4 per lane, V <= 256 * NIT).
template <bool VEC4, int NIT>
__global__ __launch_bounds__(256) void k_joint_prep(const float* __restrict__ f, const float* __restrict__ g,
                                                    const int32_t* __restrict__ labels, JointGeom jg, Ws w) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // first kernel of a forward: start the validation word and counters
    w.hdr->flags = 0;
    w.hdr->n_fallback = 0;
    w.hdr->n_partial = 0;
    w.hdr->n_count = 0;
    w.hdr->n_items_ready = 0;
    w.hdr->items_ctr = 0ull;
    w.hdr->rmw_next = 0u;
    w.hdr->partial_off = w.partial_off;
    w.hdr->count_off = w.count_off;
    w.hdr->n_cells = 0;
    w.hdr->n_pruned = 0;
  }
  const int lane = threadIdx.x & 63;
  const int grp = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int ngf = jg.B * (w.Tp / 4), ngg = jg.B * (w.U1p / 4);
  if (grp >= ngf + ngg) return;
  const bool isf = grp < ngf;
  const int R = isf ? w.Tp : w.U1p, real = isf ? jg.maxT : jg.maxU1;
  const int r0 = (isf ? grp : grp - ngf) * 4;
  const int b = r0 / R, rr0 = r0 - b * R;
  const int V = jg.V, chunks = w.Vp >> 3;
  const int q = lane >> 4, cc = (lane >> 1) & 7, e4 = (lane & 1) * 4;
  const int row = rr0 + q;
  const bool row_ok = row < real;
  const int rowc = min(row, real - 1);
  const float* src = isf ? f + (size_t)b * jg.fstride + (size_t)rowc * V : g + ((size_t)b * jg.maxU1 + rowc) * V;
  auto load4 = [&](int v, float (&x)[4]) {  // elements v..v+3 of the row, -inf past its end
    if (VEC4) {
      const f4v t = ld4(src + min(v, V - 4));
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = v < V ? t[i] : -INFINITY;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float t = src[min(v + i, V - 1)];
        x[i] = v + i < V ? t : -INFINITY;
      }
    }
  };
  constexpr int NKEEP = NIT > 0 ? NIT : 1;
  float keep[NKEEP][4];
  float m = -INFINITY;
  if (NIT > 0) {
#pragma unroll
    for (int it = 0; it < NKEEP; ++it) {
      load4((cc + 8 * it) * 8 + e4, keep[it]);  // chunks past the padded row load as -inf
      m = fmaxf(fmaxf(m, fmaxf(keep[it][0], keep[it][1])), fmaxf(keep[it][2], keep[it][3]));
    }
  } else {
    for (int c = cc; c < chunks; c += 8) {
      float x[4];
      load4(c * 8 + e4, x);
      m = fmaxf(fmaxf(m, fmaxf(x[0], x[1])), fmaxf(x[2], x[3]));
    }
  }
  m = group_max<16>(m) * kLog2e;  // the 16 lanes of a row are one DPP row
  if ((lane & 15) == 0 && row_ok) {
    // besides the maximum: the blank entry of the row and, for g, the entry of the position's next label, so that
    // k_joint_gather need not fetch them from a different 4*V-byte row for every cell
    const float xb = __builtin_fmaf(src[jg.blank], kLog2e, -m);
    if (isf) {
      w.mf2[b * jg.maxT + row] = m;
      w.fblank[b * jg.maxT + row] = xb;
    } else {
      int lab = row < jg.maxU1 - 1 ? labels[(size_t)b * (jg.maxU1 - 1) + row] : 0;
      lab = (lab < 0 || lab >= V) ? 0 : lab;
      w.mg2[b * jg.maxU1 + row] = m;
      w.gblank[b * jg.maxU1 + row] = xb;
      w.glabel[b * jg.maxU1 + row] = __builtin_fmaf(src[lab], kLog2e, -m);
    }
  }
  float* out = (isf ? w.Fp : w.Gp) + (size_t)b * R * w.Vp + tiled_off(row, e4, chunks);
  auto emit = [&](int c, const float (&x)[4]) {
    f4v y;
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = row_ok ? fast_exp2(__builtin_fmaf(x[i], kLog2e, -m)) : 0.f;  // 2^-inf = 0
    *reinterpret_cast<f4v*>(out + ((size_t)c << 8)) = y;
  };
  if (NIT > 0) {
#pragma unroll
    for (int it = 0; it < NKEEP; ++it)
      if (cc + 8 * it < chunks) emit(cc + 8 * it, keep[it]);
  } else {
    for (int c = cc; c < chunks; c += 8) {
      float x[4];
      load4(c * 8 + e4, x);
      emit(c, x);
    }
  }
}

// One thread per (b, t, u).  A lattice cell gets flab(t,u) = f2(t, y_{u+1}) (f2 = f*log2e - mf2), the one part of its
// record that has to be fetched per cell; k_joint_lse_mm puts the record together.  Entries that are no lattice
// cell, and the guard frames, get their final "log 0" record here.
__global__ __launch_bounds__(256) void k_joint_gather(const float* __restrict__ f,
                                                      const int32_t* __restrict__ labels,
                                                      const int32_t* __restrict__ in_len,
                                                      const int32_t* __restrict__ lab_len, JointGeom jg, Ws w) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int per_b = jg.maxT * jg.maxU1;
  if (i >= (long long)jg.B * per_b) return;
  const int b = (int)(i / per_b);
  const int rem = (int)(i - (long long)b * per_b);
  const int t = rem / jg.maxU1, u = rem - t * jg.maxU1;
  int T, U;
  utt_dims(in_len, lab_len, b, jg.maxT, jg.maxU1, T, U);
  const float4 neg = make_float4(kNegF, kNegF, 0.f, 0.f);
  if (t < T && u <= U) {
    float fl = 0.f;
    if (u < U) {
      int lab = labels[(size_t)b * (jg.maxU1 - 1) + u];
      if (lab < 0 || lab >= jg.V) {
        atomicOr(&w.hdr->flags, RNNT_MI355X_FLAG_BAD_LABEL);
        lab = 0;
      }
      fl = __builtin_fmaf(f[(size_t)b * jg.fstride + (size_t)t * jg.V + lab], kLog2e, -w.mf2[b * jg.maxT + t]);
    }
    w.flab[i] = fl;
  } else {
    w.rec[((size_t)b * (jg.maxT + 2) + 1 + t) * jg.maxU1 + u] = neg;
  }
  if (t == 0) w.rec[((size_t)b * (jg.maxT + 2)) * jg.maxU1 + u] = neg;
  if (t == jg.maxT - 1) w.rec[((size_t)b * (jg.maxT + 2) + 1 + jg.maxT) * jg.maxU1 + u] = neg;
}

// log2 sum_v 2^(f2(t,v) + g2(u,v)) with the exact maximum taken out first; all 64 lanes cooperate.
__device__ __noinline__ float joint_exact_ls(const float* __restrict__ f, const float* __restrict__ g,
                                             const JointGeom& jg, const Ws& w, int b, int t, int u, int lane) {
  const float* fr = f + (size_t)b * jg.fstride + (size_t)t * jg.V;
  const float* gr = g + ((size_t)b * jg.maxU1 + u) * jg.V;
  const float mf2 = w.mf2[b * jg.maxT + t], mg2 = w.mg2[b * jg.maxU1 + u];
  float m = -INFINITY;
  for (int v = lane; v < jg.V; v += 64)
    m = fmaxf(m, __builtin_fmaf(fr[v], kLog2e, -mf2) + __builtin_fmaf(gr[v], kLog2e, -mg2));
  m = group_max<64>(m);
  float s = 0.f;
  for (int v = lane; v < jg.V; v += 64)
    s += fast_exp2(__builtin_fmaf(fr[v], kLog2e, -mf2) + __builtin_fmaf(gr[v], kLog2e, -mg2) - m);
  return m + fast_log2(group_sum<64>(s));
}

// Tile coordinates of one wavefront of a batched contraction: M tiles of 32 rows vary fastest (the four waves
// of a block then share their N strip), then N strips, then utterances.
struct MmTile {
  int b, m0, n0;
  bool ok;
};
__device__ __forceinline__ MmTile mm_tile(int B, int m_tiles, int n_strips, int n_strip_cols) {
  MmTile mt;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int per_b = m_tiles * n_strips;
  mt.ok = wid < B * per_b;
  mt.b = wid / per_b;
  const int r = wid - mt.b * per_b;
  const int in = r / m_tiles;
  mt.m0 = (r - in * m_tiles) * 32;
  mt.n0 = in * n_strip_cols;
  return mt;
}

// Row of C/D register r of a 32x32 MFMA tile held by this lane (the column is lane % 32).
__device__ __forceinline__ int mm_row(int r, int lane) { return 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3); }

// S(t,u) = sum_v F(t,v) G(u,v) for a 32 x 32*NT tile per wavefront, then the cell records.  Both operands are
// K-contiguous: a lane loads 4 consecutive v of its row per step (lanes 0-31 take v = k..k+3, lanes 32-63
// v = k+4..k+7; the order of the sum over v is free) and feeds them to 4 MFMAs.  The operands of the next
// step are in flight during this step's MFMAs.
template <int NT>
__global__ __launch_bounds__(256) void k_joint_lse_mm(const float* __restrict__ f, const float* __restrict__ g,
                                                      const int32_t* __restrict__ in_len,
                                                      const int32_t* __restrict__ lab_len, JointGeom jg, Ws w) {
  const int lane = threadIdx.x & 63, l32 = lane & 31, h = lane >> 5;
  const MmTile mt = mm_tile(jg.B, w.Tp / 32, (w.U1p + 32 * NT - 1) / (32 * NT), 32 * NT);
  if (!mt.ok) return;
  const int b = mt.b, t0 = mt.m0, u0 = mt.n0;
  int T, U;
  utt_dims(in_len, lab_len, b, jg.maxT, jg.maxU1, T, U);
  if (t0 >= T || u0 > U) return;  // no lattice cell in this tile: its records already say "log 0"
  const int Vp = w.Vp, chunks = Vp >> 3;
  const float* ap = w.Fp + (size_t)b * w.Tp * Vp + tiled_off(t0 + l32, 4 * h, chunks);
  const float* bp[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)  // a strip may reach past the padded rows of G: such tiles re-read the last one
    bp[j] = w.Gp + (size_t)b * w.U1p * Vp + tiled_off(min(u0 + 32 * j, w.U1p - 32) + l32, 4 * h, chunks);
  f16v acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // Two register sets of two column chunks (16 v) each, used alternately: the loads of one set are in flight
  // during the 8*NT MFMAs of the other.  (Four sets in flight measured slower: fewer resident waves.)
  const int nsets = chunks >> 1;
  f4v as[kMmSets][2], bs[kMmSets][NT][2];
  auto load_set = [&](int set, f4v (&a)[2], f4v (&bq)[NT][2]) {
    const size_t o = (size_t)min(set, nsets - 1) << 9;  // past the end: re-reads the last set, unused
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      a[s2] = ld4(ap + o + ((size_t)s2 << 8));
#pragma unroll
      for (int j = 0; j < NT; ++j) bq[j][s2] = ld4(bp[j] + o + ((size_t)s2 << 8));
    }
  };
  auto mfma_set = [&](const f4v (&a)[2], const f4v (&bq)[NT][2]) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2][i], bq[j][s2][i], acc[j], 0, 0, 0);
  };
  load_set(0, as[0], bs[0]);
  for (int s0 = 0; s0 < nsets; s0 += 2) {  // nsets is even; no branch inside the body, so the loads stay where they are
    load_set(s0 + 1, as[1], bs[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfma_set(as[0], bs[0]);
    __builtin_amdgcn_sched_barrier(0);
    load_set(s0 + 2, as[0], bs[0]);
    __builtin_amdgcn_sched_barrier(0);
    mfma_set(as[1], bs[1]);
    __builtin_amdgcn_sched_barrier(0);
  }
  // Records of the tile's cells: {f2(t,blank) + g2(u,blank) - ls, f2(t,y) + g2(u,y) - ls, ls, flag} with
  // ls = log2 S.  The per-cell parts are all requested before the first record is stored (lanes without a cell
  // re-read a neighbouring one).  Cells whose sum underflowed are left for the slow loop below.
  float4* frame0 = w.rec + ((size_t)b * (jg.maxT + 2) + 1) * jg.maxU1;
  const float* fbl = w.fblank + (size_t)b * jg.maxT;
  const float* flb = w.flab + (size_t)b * jg.maxT * jg.maxU1;
  float fbv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) fbv[r] = fbl[min(t0 + mm_row(r, lane), T - 1)];
  unsigned long long badmask = 0;  // bit 16*j + r
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int u = u0 + 32 * j + l32;
    const int uc = min(u, U);
    const float gb = w.gblank[b * jg.maxU1 + uc], gl = w.glabel[b * jg.maxU1 + uc];
    float flv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) flv[r] = flb[(size_t)min(t0 + mm_row(r, lane), T - 1) * jg.maxU1 + uc];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = t0 + mm_row(r, lane);
      const bool valid = t < T && u <= U;
      const float S = acc[j][r];
      const bool bad = valid && !(S >= kJointMinSum);
      badmask |= bad ? (1ull << (16 * j + r)) : 0ull;
      if (valid && !bad) {
        const float ls = fast_log2(S);
        frame0[(size_t)t * jg.maxU1 + u] =
            make_float4((fbv[r] + gb) - ls, (u < U) ? (flv[r] + gl) - ls : kNegF, ls, 0.f);
      }
    }
  }
  if (__ballot(badmask != 0)) {  // rare: redo those cells with the exact maximum, one at a time, all lanes helping
    int n_bad = 0;
    for (int q = 0; q < 16 * NT; ++q) {
      const int j = q >> 4, r = q & 15;
      const int t = t0 + mm_row(r, lane), u = u0 + 32 * j + l32;
      const bool bad = (badmask >> q) & 1ull;
      unsigned long long m = __ballot(bad);
      float ls = 0.f;
      while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const float e = joint_exact_ls(f, g, jg, w, b, __shfl(t, src), __shfl(u, src), lane);
        if (lane == src) ls = e;
      }
      if (bad) {
        const float xb = fbl[t] + w.gblank[b * jg.maxU1 + u];
        const float xl = flb[(size_t)t * jg.maxU1 + u] + w.glabel[b * jg.maxU1 + u];
        frame0[(size_t)t * jg.maxU1 + u] = make_float4(xb - ls, (u < U) ? xl - ls : kNegF, ls, 1.f);
        ++n_bad;
      }
    }
    if (n_bad) atomicAdd(&w.hdr->n_fallback, n_bad);
  }
}

// One block per (b, 8 frames, 32 label positions) of the padded lattice, one thread per cell: {c, s*pb, s*pl,
// label code} (c = log2 gamma - log2 S, -inf when the cell is pruned), the sums of s*pb and s*pl over the block's 8
// frames, and C = s * 2^c for the contractions (0 where
// pruned, flagged, or outside the lattice), written in both tiled orientations (the transposed one through LDS so
// that both leave as runs of 32 bytes).
__global__ __launch_bounds__(256) void k_joint_coef(const int32_t* __restrict__ labels,
                                                    const int32_t* __restrict__ in_len,
                                                    const int32_t* __restrict__ lab_len,
                                                    const float* __restrict__ grad_scale, JointGeom jg, Ws w) {
  __shared__ float tile[8][33];
  __shared__ float colp[2][8][33];
  const int ub = w.U1p >> 5, tb = w.Tp >> 3;
  const int b = blockIdx.x / (ub * tb);
  const int r = blockIdx.x - b * ub * tb;
  const int t0 = (r / ub) * 8, u0 = (r - (r / ub) * ub) * 32;
  const int tt = threadIdx.x >> 5, uu = threadIdx.x & 31;
  const int t = t0 + tt, u = u0 + uu;
  float cm = 0.f, pbv = 0.f, plv = 0.f;
  if (t < jg.maxT && u < jg.maxU1) {
    int T, U;
    utt_dims(in_len, lab_len, b, jg.maxT, jg.maxU1, T, U);
    const CellRaw raw = cell_load(w, labels, b, t, u, T, U, jg.maxT, jg.maxU1, jg.V, true);
    float4 cf = cell_coef<false>(raw, jg.prune_log2, jg.blank);  // (the record's fourth entry is this path's fallback flag)
    const float sc = grad_scale ? grad_scale[b] : 1.f;
    cf.y *= sc;
    cf.z *= sc;
    const bool live = cf.x > -INFINITY;
    const bool flagged = live && raw.rc.w != 0.f;
    cf.w = __int_as_float((raw.lab + 1) | (flagged ? kJointFlagBit : 0));
    if (live && !flagged) cm = sc * fast_exp2(cf.x);
    pbv = cf.y;
    plv = cf.z;
    w.coef[((size_t)b * jg.maxT + t) * jg.maxU1 + u] = cf;
  }
  // the 32 positions of a frame in this block are one half wavefront: their blank terms, summed
  const float pbsum = group_sum<32>(pbv);
  if (uu == 0) w.rowpart[((size_t)b * w.Tp + t) * ub + (u0 >> 5)] = pbsum;
  const size_t per_b = (size_t)w.Tp * w.U1p;
  w.Ctu[b * per_b + tiled_off(t, u, w.U1p >> 3)] = cm;
  tile[tt][uu] = cm;
  colp[0][tt][uu] = pbv;
  colp[1][tt][uu] = plv;
  __syncthreads();
  const int u2 = threadIdx.x >> 3, t2 = threadIdx.x & 7;
  w.Cut[b * per_b + tiled_off(u0 + u2, t0 + t2, w.Tp >> 3)] = tile[t2][u2];
  // the block's 8 frames, summed per label position: k_joint_corr_g adds these up instead of re-reading every cell
  if (threadIdx.x < 32) {
    float sb = 0.f, sl = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sb += colp[0][i][uu];
      sl += colp[1][i][uu];
    }
    w.colpart[((size_t)b * tb + (t0 >> 3)) * w.U1p + u] = make_float2(sb, sl);
  }
}

// NT consecutive floats
template <int NT>
struct FVec;
template <>
struct FVec<1> { typedef float type; };
template <>
struct FVec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <>
struct FVec<4> { typedef float type __attribute__((ext_vector_type(4))); };
template <int NT>
__device__ __forceinline__ float fvec_get(const typename FVec<NT>::type& v, int j) { return v[j]; }
template <>
__device__ __forceinline__ float fvec_get<1>(const float& v, int) { return v; }

// out(m, n) = E(m, n) * sum_k A(m, k) Bm(k, n) for one wavefront tile of 32 rows x 32*NT columns, where
//   A   is K-contiguous tiled (`a_base` + row tile m0, `a_chunks` = padded K / 8),
//   Bm  is N-contiguous tiled with rows k (`b_base`, chunks = Vp / 8): G for d f, F for d g,
//   E   is the tiled matrix multiplied in element-wise at the end (`e_base`, rows m): F for d f, G for d g.
// Lane l of MFMA tile j owns column n0 + NT*(l%32) + j, so that one load of NT consecutive floats feeds all NT
// tiles and the results leave as NT consecutive floats.
// What the epilogue of d f subtracts: d f(t, y) -= sum over the positions u whose next label is y of s*pl(t,u),
// d f(t, blank) -= sum_u s*pb(t,u).  Positions that share a target are linked into a chain (k_joint_labelmap);
// the lane that owns column y walks it in order -- no atomics, no second pass over d f.
struct DfCorr {
  const int* head;       // [Vp] of this utterance
  const int* nxt;        // [maxU1 + 2]
  const float4* coef;    // [maxT][maxU1] of this utterance
  const float* rowpart;  // [Tp][nub]
  int maxU1, nub, blank_pos;  // blank_pos = U_b + 1: the chain entry that stands for the blank
};

// Epilogue of a gradient contraction: the element-wise factor, the blank / label chains (CORR), the stores.
template <int NT, bool CORR>
__device__ __forceinline__ void joint_grad_epilogue(f16v (&acc)[NT], const int (&hd)[NT], const float* __restrict__ e_base,
                                                    int Vp, int m0, int col, int m_real, int V, float* __restrict__ out,
                                                    bool vec_ok, int lane, const DfCorr& dc) {
  typedef typename FVec<NT>::type vecn;
  const int chunks = Vp >> 3;
  // all 16 element-wise factors are loaded before the first store (rows past the end re-read the last one);
  // requesting them ahead of the K loop measured slower (64 more registers, fewer resident waves)
  vecn e[16];
#pragma unroll
  for (int r = 0; r < 16; ++r)
    e[r] = *reinterpret_cast<const vecn*>(e_base + tiled_off(min(m0 + mm_row(r, lane), m_real - 1), col, chunks));
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] *= fvec_get<NT>(e[r], j);
  if (CORR) {
    // The chain of positions whose target is this lane's column: its first kChain entries are fetched for all 16
    // rows at once (independent loads, other lanes re-read entry 0 and discard it); longer chains, which need
    // more than kChain repeats of one symbol, take the serial loop.
    constexpr int kChain = 2;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      int k = hd[j];
#pragma unroll
      for (int i = 0; i < kChain; ++i) {
        const int kc = k;
        k = kc >= 0 ? dc.nxt[kc] : -1;
        if (__ballot(kc >= 0) == 0ull) continue;  // uniform
        const bool is_blank = kc == dc.blank_pos;
        const bool is_label = kc >= 0 && !is_blank;
        if (__ballot(is_label) != 0ull) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = min(m0 + mm_row(r, lane), m_real - 1);
            const float x = dc.coef[(size_t)m * dc.maxU1 + (is_label ? kc : 0)].z;
            acc[j][r] -= is_label ? x : 0.f;
          }
        }
        if (__ballot(is_blank) != 0ull) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = min(m0 + mm_row(r, lane), m_real - 1);
            float rs = 0.f;
            for (int q = 0; q < dc.nub; ++q) rs += dc.rowpart[(size_t)m * dc.nub + q];
            acc[j][r] -= is_blank ? rs : 0.f;
          }
        }
      }
      if (__ballot(k >= 0) != 0ull) {  // rare
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = min(m0 + mm_row(r, lane), m_real - 1);
          double corr = 0.0;  // float64: see k_joint_corr_f
          for (int kk = k; kk >= 0; kk = dc.nxt[kk]) {
            if (kk == dc.blank_pos) {
              for (int q = 0; q < dc.nub; ++q) corr += (double)dc.rowpart[(size_t)m * dc.nub + q];
            } else {
              corr += (double)dc.coef[(size_t)m * dc.maxU1 + kk].z;
            }
          }
          acc[j][r] -= (float)corr;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + mm_row(r, lane);
    float* o = out + (size_t)m * V + col;
    if (m < m_real) {
      if (vec_ok && col + NT <= V) {
        vecn res;
        if constexpr (NT == 1) res = acc[0][r];
        else {
#pragma unroll
          for (int j = 0; j < NT; ++j) res[j] = acc[j][r];
        }
        *reinterpret_cast<vecn*>(o) = res;
      } else {
#pragma unroll
        for (int j = 0; j < NT; ++j)
          if (col + j < V) o[j] = acc[j][r];
      }
    }
  }
}

template <int NT, bool CORR>
__device__ __forceinline__ void joint_grad_mm(const float* __restrict__ a_base, int a_chunks, int kbeg, int kmax,
                                              const float* __restrict__ b_base, const float* __restrict__ e_base,
                                              int Vp, int m0, int n0, int m_real, int V, float* __restrict__ out,
                                              bool vec_ok, int lane, const DfCorr& dc) {
  typedef typename FVec<NT>::type vecn;
  const int l32 = lane & 31, h = lane >> 5, chunks = Vp >> 3;
  const int col = n0 + NT * l32;
  int hd[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) hd[j] = CORR ? dc.head[col + j] : -1;  // col + j < Vp
  const float* ap = a_base + tiled_off(m0 + l32, 4 * h, a_chunks);
  const float* bp = b_base + tiled_off(4 * h, col, chunks);  // + 8 floats per row, + 256*chunks per 32 rows
  f16v acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  auto b_off = [&](int k0, int i) { return ((size_t)(k0 >> 5) * chunks << 8) + (((k0 & 31) + i) << 3); };
  // A ring of D register sets (8 columns of A and 8 rows of Bm each): a set is re-loaded right after the 4*NT MFMAs
  // that consumed it, so its loads have the MFMAs of the other D-1 sets to come back.  (Written as "load the next
  // set, then multiply this one" with two sets, the compiler moved the loads below the MFMAs they were meant to
  // overlap; a load that follows the last use of its own registers cannot be moved up, and the scheduling barrier
  // keeps it from moving down.)
#ifndef RNNT_JOINT_RING
#define RNNT_JOINT_RING 3
#endif
  constexpr int D = (NT == 4 && CORR) ? 2 : RNNT_JOINT_RING;  // (the epilogue of that form needs the registers)
  f4v a[D];
  vecn bv[D][4];
  auto load_set = [&](int k0, f4v& av, vecn (&bw)[4]) {
    av = ld4(ap + ((size_t)(k0 >> 3) << 8));
#pragma unroll
    for (int i = 0; i < 4; ++i) bw[i] = *reinterpret_cast<const vecn*>(bp + b_off(k0, i));
  };
  auto mfma_set = [&](const f4v& av, const vecn (&bw)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], fvec_get<NT>(bw[i], j), acc[j], 0, 0, 0);
  };
  if (kmax > kbeg) {  // K range [kbeg, kmax), both multiples of 8
    const int klast = kmax - 8;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      load_set(min(kbeg + 8 * d, klast), a[d], bv[d]);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int k0 = kbeg; k0 < kmax; k0 += 8 * D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (k0 + 8 * d < kmax) mfma_set(a[d], bv[d]);  // uniform
        load_set(min(k0 + 8 * (d + D), klast), a[d], bv[d]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  joint_grad_epilogue<NT, CORR>(acc, hd, e_base, Vp, m0, col, m_real, V, out, vec_ok, lane, dc);
}

// d f(t,v) = F(t,v) * sum_u C(t,u) G(u,v): M = frames, N = vocabulary, K = label positions.
// CORR: the blank / label terms are subtracted in the epilogue (k_joint_labelmap has run); otherwise
// k_joint_corr_f follows.
// (compiled for two resident waves per SIMD: the epilogue with the label chains came to 194 + 64 registers, two more
// than that allows, and ran one wave per SIMD)
template <int NT, bool CORR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_joint_df_mm(float* __restrict__ df, const int32_t* __restrict__ in_len,
                                                     const int32_t* __restrict__ lab_len, JointGeom jg, Ws w,
                                                     bool vec_ok) {
  const MmTile mt = mm_tile(jg.B, w.Tp / 32, w.Vp / (32 * NT), 32 * NT);
  if (!mt.ok) return;
  int T, U;
  utt_dims(in_len, lab_len, mt.b, jg.maxT, jg.maxU1, T, U);
  const int kmax = (mt.m0 < T) ? (U + 8) & ~7 : 0;  // C is zero beyond column U_b and in frames >= T_b
  const size_t cb = (size_t)mt.b * w.Tp * w.U1p;
  DfCorr dc;
  dc.head = w.head_of + (size_t)mt.b * w.Vp;
  dc.nxt = w.nxt + (size_t)mt.b * (jg.maxU1 + 2);
  dc.coef = w.coef + (size_t)mt.b * jg.maxT * jg.maxU1;
  dc.rowpart = w.rowpart + (size_t)mt.b * w.Tp * (w.U1p >> 5);
  dc.maxU1 = jg.maxU1; dc.nub = w.U1p >> 5; dc.blank_pos = U + 1;
  joint_grad_mm<NT, CORR>(w.Ctu + cb, w.U1p >> 3, 0, kmax, w.Gp + (size_t)mt.b * w.U1p * w.Vp,
                          w.Fp + (size_t)mt.b * w.Tp * w.Vp, w.Vp, mt.m0, mt.n0, jg.maxT, jg.V,
                          df + (size_t)mt.b * jg.dfstride, vec_ok, threadIdx.x & 63, dc);
}

// d g(u,v) = G(u,v) * sum_t C(t,u) F(t,v): M = label positions, N = vocabulary, K = frames.  With w.dg_split > 1
// the frames are cut into that many slices, each wavefront tile takes one slice and writes its partial result to
// w.dgpart; k_joint_corr_g adds the slices up.
template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_joint_dg_mm(float* __restrict__ dg, const int32_t* __restrict__ in_len,
                                                     const int32_t* __restrict__ lab_len, JointGeom jg, Ws w,
                                                     bool vec_ok) {
  const int ks = w.dg_split;
  const MmTile mt = mm_tile(jg.B * ks, w.U1p / 32, w.Vp / (32 * NT), 32 * NT);
  if (!mt.ok) return;
  const int slice_i = mt.b / jg.B, b = mt.b - slice_i * jg.B;  // slices vary slowest
  int T, U;
  utt_dims(in_len, lab_len, b, jg.maxT, jg.maxU1, T, U);
  const int kmax = (mt.m0 <= U) ? (T + 7) & ~7 : 0;  // C is zero in frames >= T_b and columns > U_b
  const int slice = (kmax + 16 * ks - 1) / (16 * ks) * 16;
  const int kbeg = min(kmax, slice_i * slice), kend = min(kmax, kbeg + slice);
  const size_t cb = (size_t)b * w.Tp * w.U1p;
  float* out = ks > 1 ? w.dgpart + ((size_t)slice_i * jg.B + b) * jg.maxU1 * jg.V : dg + (size_t)b * jg.maxU1 * jg.V;
  joint_grad_mm<NT, false>(w.Cut + cb, w.Tp >> 3, kbeg, kend, w.Fp + (size_t)b * w.Tp * w.Vp,
                           w.Gp + (size_t)b * w.U1p * w.Vp, w.Vp, mt.m0, mt.n0, jg.maxU1, jg.V, out, vec_ok,
                           threadIdx.x & 63, DfCorr{});
}

// ---------------------------------------------------------------------------
// The d g contraction with the N-contiguous operand shared by a block through LDS (strips of 128 columns).
// Per step of 8 along K a wavefront of the register-fed form loads 1 KiB of A and 4 KiB of Bm for 16 matrix
// instructions, and the four wavefronts of a compute unit's SIMDs ask the path from L2 for more bytes than 256
// cycles of matrix work leave time for.  Here a block is four wavefronts on four CONSECUTIVE row tiles of ONE
// column strip: Bm's 8 x 128 piece of a step is fetched once per block (a float4
// per thread, two steps ahead, through registers into one of two LDS buffers, one barrier per step) and read by
// each wavefront with four 16-byte LDS loads; A stays per wavefront, straight from memory, in a ring of three.
// LDS layout of a piece: [16 chunks of 8 columns][8 rows][8 columns], chunks 68 floats apart (the 16 lanes pairs of
// a wavefront that read the same row of different chunks would otherwise meet in one bank).
// Measured at B=64, T=400, U=100, V=1024: d g (50 steps along the frames, 4 row tiles = one full block) 76.5 -> 72.1
// us; the same for d f (13 steps along the label positions, an epilogue as large as the loop, 13 row tiles = every
// fourth block with three idle wavefronts) 121.6 -> 146.2 us: d f keeps the register-fed form, and so does d g
// when its row tiles do not fill whole blocks (U+1 = 151: five tiles).
// ---------------------------------------------------------------------------
constexpr int kMmLdsChunk = 68;                      // floats between two column chunks of a staged piece
constexpr int kMmLdsPiece = 16 * kMmLdsChunk;        // floats per staged piece (8 x 128 of Bm)

// K range [kbeg, kmax) is the same for the whole block; `live` = this wavefront's row tile exists.
__device__ __forceinline__ void joint_mm_loop_lds(f16v (&acc)[4], const float* __restrict__ ap, int kbeg, int kmax,
                                                  const float* __restrict__ b_strip, int chunks, float* __restrict__ lds,
                                                  bool live, int lane) {
  const int l32 = lane & 31, h = lane >> 5;
  const int tid = threadIdx.x;
  // this thread's float4 of a piece: chunk tid / 16, float4 tid % 16 of the chunk's 64 floats (row (tid % 16) / 2)
  const float* bsrc = b_strip + ((size_t)(tid >> 4) << 8) + ((tid & 15) << 2);
  float* bdst = lds + (tid >> 4) * kMmLdsChunk + ((tid & 15) << 2);
  const float* brd = lds + (l32 >> 1) * kMmLdsChunk + (4 * h) * 8 + 4 * (l32 & 1);
  auto b_piece = [&](int k0) { return ld4(bsrc + (((size_t)(k0 >> 5) * chunks) << 8) + ((k0 & 31) << 3)); };
  const int nsteps = (kmax - kbeg) >> 3;
  if (nsteps <= 0) return;  // (block-uniform)
  const int klast = kmax - 8;
  f4v a[3];
  f4v stage[2];
#pragma unroll
  for (int d = 0; d < 3; ++d) a[d] = ld4(ap + ((size_t)(min(kbeg + 8 * d, klast) >> 3) << 8));
  {
    const f4v first = b_piece(kbeg);
    stage[1] = b_piece(min(kbeg + 8, klast));
    *reinterpret_cast<f4v*>(bdst) = first;
  }
  __syncthreads();
  // step s reads LDS buffer s % 2; stage[s % 2] receives the piece of step s + 2, stage[(s + 1) % 2] (the piece of
  // step s + 1, requested a step ago) goes into buffer (s + 1) % 2 at the end of step s
  for (int s0 = 0; s0 < nsteps; s0 += 6) {  // 6 = lcm of the ring of A (3) and the two buffers
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      const int s = s0 + d;
      if (s < nsteps) {  // (block-uniform)
        const int k0 = kbeg + 8 * s;
        stage[d & 1] = b_piece(min(k0 + 16, klast));
        const float* rd = brd + (d & 1) * kMmLdsPiece;
        f4v bw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bw[i] = *reinterpret_cast<const f4v*>(rd + i * 8);
        if (live) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[d % 3][i], bw[i][j], acc[j], 0, 0, 0);
        }
        a[d % 3] = ld4(ap + ((size_t)(min(k0 + 24, klast) >> 3) << 8));
        *reinterpret_cast<f4v*>(bdst + ((d + 1) & 1) * kMmLdsPiece) = stage[(d + 1) & 1];
        __syncthreads();
      }
    }
  }
}

// Block -> (utterance, column strip, group of four row tiles); wavefront w takes row tile 4 * group + w.
struct MmBlockTile {
  int b, strip, m_first, m0;  // m_first: first row of the group; m0: this wavefront's (clamped to the last tile)
  bool live;
};
__device__ __forceinline__ MmBlockTile mm_block_tile(int m_tiles, int strips) {
  MmBlockTile t;
  const int groups = (m_tiles + 3) >> 2;
  int id = blockIdx.x;
  const int grp = id % groups;
  id /= groups;
  t.strip = id % strips;
  t.b = id / strips;
  const int mine = 4 * grp + (int)(threadIdx.x >> 6);
  t.live = mine < m_tiles;
  t.m_first = 128 * grp;
  t.m0 = 32 * min(mine, m_tiles - 1);
  return t;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_joint_dg_mm_lds(
    float* __restrict__ dg, const int32_t* __restrict__ in_len, const int32_t* __restrict__ lab_len, JointGeom jg, Ws w,
    bool vec_ok) {
  __shared__ __attribute__((aligned(16))) float lds[2 * kMmLdsPiece];
  const int ks = w.dg_split;
  MmBlockTile t = mm_block_tile(w.U1p / 32, w.Vp / 128);  // t.b counts (slice, utterance) pairs, slices slowest
  const int slice_i = t.b / jg.B, b = t.b - slice_i * jg.B;
  const int lane = threadIdx.x & 63, l32 = lane & 31, h = lane >> 5;
  int T, U;
  utt_dims(in_len, lab_len, b, jg.maxT, jg.maxU1, T, U);
  const int kmax = (t.m_first <= U) ? (T + 7) & ~7 : 0;  // C is zero in frames >= T_b and columns > U_b
  const int slice = (kmax + 16 * ks - 1) / (16 * ks) * 16;
  const int kbeg = min(kmax, slice_i * slice), kend = min(kmax, kbeg + slice);
  const int chunks = w.Vp >> 3, n0 = 128 * t.strip, col = n0 + 4 * l32;
  int hd[4] = {-1, -1, -1, -1};
  f16v acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const float* a_base = w.Cut + (size_t)b * w.Tp * w.U1p;
  const float* b_base = w.Fp + (size_t)b * w.Tp * w.Vp;
  joint_mm_loop_lds(acc, a_base + tiled_off(t.m0 + l32, 4 * h, w.Tp >> 3), kbeg, kend, b_base + ((size_t)(n0 >> 3) << 8), chunks,
                    lds, t.live, lane);
  if (!t.live) return;
  float* out = ks > 1 ? w.dgpart + ((size_t)slice_i * jg.B + b) * jg.maxU1 * jg.V : dg + (size_t)b * jg.maxU1 * jg.V;
  joint_grad_epilogue<4, false>(acc, hd, w.Gp + (size_t)b * w.U1p * w.Vp, w.Vp, t.m0, col, jg.maxU1, jg.V, out, vec_ok, lane,
                                DfCorr{});
}

// Blank and label terms of d f as a pass of their own (small vocabularies, where labels repeat so often that the
// chains are long and d f is small).  d f(t, y) -= sum over the positions u whose next label is y of s*pl(t,u), and
// d f(t, blank) -= sum_u s*pb(t,u).  One block per (utterance, fb frames): it links the positions that share a
// target (labels repeat; the blank is one more target) into chains, then one thread per (frame, first position of
// a chain) adds the chain up in order and subtracts it once -- no atomics, fixed order.
__global__ __launch_bounds__(256) void k_joint_corr_f(float* __restrict__ df, const int32_t* __restrict__ labels,
                                                      const int32_t* __restrict__ in_len,
                                                      const int32_t* __restrict__ lab_len, JointGeom jg, Ws w,
                                                      int fb) {
  extern __shared__ int jcs[];  // tgt[n], next[n], head[n], then val[fb][n]   (n = maxU1 + 1)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n = jg.maxU1 + 1;
  int* tgt = jcs;
  int* nxt = jcs + n;
  int* head = jcs + 2 * n;
  float* val = reinterpret_cast<float*>(jcs + 3 * n);
  const int tblocks = (jg.maxT + fb - 1) / fb;
  const int b = blockIdx.x / tblocks, t0 = (blockIdx.x - b * tblocks) * fb;
  int T, U;
  utt_dims(in_len, lab_len, b, jg.maxT, jg.maxU1, T, U);
  if (t0 >= T) return;  // uniform over the block
  // the label terms of the block's frames (all requested before anything waits), blank sums at U+1
  const int nrow = min(fb, T - t0);
  for (int i = wv; i < nrow; i += 4) {
    const float4* cfrow = w.coef + ((size_t)b * jg.maxT + t0 + i) * jg.maxU1;
    float* vr = val + (size_t)i * n;
    float sumb = 0.f;
    for (int u = lane; u <= U; u += 64) {
      const float4 cf = cfrow[u];
      sumb += cf.y;
      vr[u] = cf.z;
    }
    sumb = group_sum<64>(sumb);
    if (lane == 0) vr[U + 1] = sumb;
  }
  // targets: label of position u < U (clamped like everywhere else), nothing at u = U, the blank at index U+1
  for (int u = threadIdx.x; u <= U + 1; u += 256) {
    int y = -1;
    if (u < U) {
      y = labels[(size_t)b * (jg.maxU1 - 1) + u];
      if (y < 0 || y >= jg.V) y = 0;
    } else if (u == U + 1) {
      y = jg.blank;
    }
    tgt[u] = y;
  }
  __syncthreads();
  for (int u = threadIdx.x; u <= U + 1; u += 256) {
    const int y = tgt[u];
    int nx = -1, first = (y >= 0) ? 1 : 0;
    if (y >= 0) {
      for (int k = U + 1; k > u; --k) nx = (tgt[k] == y) ? k : nx;
      for (int k = 0; k < u; ++k) first = (tgt[k] == y) ? 0 : first;
    }
    nxt[u] = nx;
    head[u] = first;
  }
  __syncthreads();
  // one thread per (frame, chain head): add the chain up in order, subtract it once
  for (int idx = threadIdx.x; idx < nrow * (U + 2); idx += 256) {
    const int i = idx / (U + 2), u = idx - i * (U + 2);
    if (!head[u]) continue;
    const float* vr = val + (size_t)i * n;
    // float64: a chain can hold hundreds of nearly equal terms (a symbol repeated along the whole utterance), whose
    // float32 running sum rounds the same way at every step
    double sacc = 0.0;
    for (int k = u; k >= 0; k = nxt[k]) sacc += (double)vr[k];
    df[(size_t)b * jg.dfstride + (size_t)(t0 + i) * jg.V + tgt[u]] -= (float)sacc;
  }
}

// One block per utterance: links the label positions that share a target symbol into chains (labels repeat; the
// blank is one more target, standing at position U_b + 1) and records, per symbol, the first position of its
// chain.  The epilogue of the d f contraction walks these chains.
__global__ __launch_bounds__(256) void k_joint_labelmap(const int32_t* __restrict__ labels,
                                                        const int32_t* __restrict__ in_len,
                                                        const int32_t* __restrict__ lab_len, JointGeom jg, Ws w) {
  extern __shared__ int tgt[];  // [maxU1 + 2]
  const int b = blockIdx.x;
  int T, U;
  utt_dims(in_len, lab_len, b, jg.maxT, jg.maxU1, T, U);
  int* head = w.head_of + (size_t)b * w.Vp;
  int* nxt = w.nxt + (size_t)b * (jg.maxU1 + 2);
  for (int v = threadIdx.x; v < w.Vp; v += 256) head[v] = -1;
  // targets: label of position u < U (clamped like everywhere else), nothing at u = U, the blank at index U+1
  for (int u = threadIdx.x; u <= U + 1; u += 256) {
    int y = -1;
    if (u < U) {
      y = labels[(size_t)b * (jg.maxU1 - 1) + u];
      if (y < 0 || y >= jg.V) y = 0;
    } else if (u == U + 1) {
      y = jg.blank;
    }
    tgt[u] = y;
  }
  __syncthreads();
  for (int u = threadIdx.x; u <= U + 1; u += 256) {
    const int y = tgt[u];
    int nx = -1;
    bool first = y >= 0;
    if (y >= 0) {
      for (int k = U + 1; k > u; --k) nx = (tgt[k] == y) ? k : nx;
      for (int k = 0; k < u; ++k) first = first && tgt[k] != y;
    }
    nxt[u] = nx;
    if (first) head[y] = u;
  }
}

// Blank and label terms of d g: 64 label positions per block, the frames split over the block's 16 waves.  When
// the contraction was cut into slices (w.dg_split > 1) the block first adds the slices of its 64 rows up, in order.
__global__ __launch_bounds__(1024) void k_joint_corr_g(float* __restrict__ dg, const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ in_len,
                                                       const int32_t* __restrict__ lab_len, JointGeom jg, Ws w) {
  __shared__ double red[2][16][64];  // float64 sums: see k_joint_corr_f
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ublocks = (jg.maxU1 + 63) / 64;
  const int b = blockIdx.x / ublocks;
  const int ub0 = (blockIdx.x - b * ublocks) * 64;
  const int u = ub0 + lane;
  if (w.dg_split > 1) {
    const int rows = min(64, jg.maxU1 - ub0);
    const size_t base = ((size_t)b * jg.maxU1 + ub0) * jg.V, per_slice = (size_t)jg.B * jg.maxU1 * jg.V;
    const int ks = w.dg_split;  // 2, 4 or 8
#pragma unroll 2
    for (int i = threadIdx.x; i < rows * jg.V; i += 1024) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = k < ks ? w.dgpart[(size_t)k * per_slice + base + i] : 0.f;  // all in flight
      float acc = v[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) acc += v[k];
      dg[base + i] = acc;
    }
  }
  int T, U;
  utt_dims(in_len, lab_len, b, jg.maxT, jg.maxU1, T, U);
  double sb = 0.0, sl = 0.0;
  if (u <= U) {
    const float2* cp = w.colpart + (size_t)b * (w.Tp >> 3) * w.U1p + u;
    for (int t8 = wv; t8 < (T + 7) >> 3; t8 += 16) {  // cells past T_b contribute zeros
      const float2 c2 = cp[(size_t)t8 * w.U1p];
      sb += (double)c2.x;
      sl += (double)c2.y;
    }
  }
  red[0][wv][lane] = sb;
  red[1][wv][lane] = sl;
  __syncthreads();  // also orders the sums of slices above before the read-modify-writes below
  if (wv == 0 && u <= U && T > 0) {
    sb = 0.0;
    sl = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      sb += red[0][i][lane];
      sl += red[1][i][lane];
    }
    float* out = dg + ((size_t)b * jg.maxU1 + u) * jg.V;
    out[jg.blank] -= (float)sb;
    if (u < U) {
      const int l = labels[(size_t)b * (jg.maxU1 - 1) + u];
      out[(l < 0 || l >= jg.V) ? 0 : l] -= (float)sl;
    }
  }
}

// Exponent-domain contribution of the flagged cells to d f: a wave owns TT frames and loops over u.
template <int KE, int TT>
__device__ __forceinline__ void joint_fix_f(int block, const float* __restrict__ f, const float* __restrict__ g,
                                            float* __restrict__ df, const int32_t* __restrict__ in_len,
                                            const int32_t* __restrict__ lab_len,
                                            const float* __restrict__ grad_scale, const JointGeom& jg, const Ws& w) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ntile = (jg.maxT + TT - 1) / TT;
  const int blocks_per_b = (ntile + 3) / 4;
  const int b = block / blocks_per_b;
  const int tile = (block - b * blocks_per_b) * 4 + wv;
  if (tile >= ntile) return;
  const int t0 = tile * TT;
  const int V = jg.V, maxU1 = jg.maxU1;
  int T, U;
  utt_dims(in_len, lab_len, b, jg.maxT, maxU1, T, U);
  const float sc = grad_scale ? grad_scale[b] : 1.f;
  const int nt = min(TT, max(T - t0, 0));
  if (nt <= 0) return;
  float acc[TT][KE];
  bool any = false;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int k = 0; k < KE; ++k) acc[tt][k] = 0.f;
  float f2[TT][KE];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int t = min(t0 + tt, jg.maxT - 1);
    jrow_load<KE>(f + (size_t)b * jg.fstride + (size_t)t * V, V, lane, w.mf2[b * jg.maxT + t], f2[tt]);
  }
  const float* gb = g + (size_t)b * maxU1 * V;
  const float* mg = w.mg2 + b * maxU1;
  for (int u = 0; u <= U; ++u) {
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      if (tt >= nt) continue;
      const float4 cf = w.coef[((size_t)b * jg.maxT + t0 + tt) * maxU1 + u];  // uniform over the wave
      if (!(__float_as_int(cf.w) & kJointFlagBit)) continue;
      any = true;
      float g2[KE];
      jrow_load<KE>(gb + (size_t)u * V, V, lane, mg[u], g2);
#pragma unroll
      for (int k = 0; k < KE; ++k) acc[tt][k] += sc * fast_exp2(f2[tt][k] + g2[k] + cf.x);
    }
  }
  if (!any) return;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    if (tt >= nt) continue;
    float* out = df + (size_t)b * jg.dfstride + (size_t)(t0 + tt) * V;
#pragma unroll
    for (int k = 0; k < KE; ++k) {
      const int v = lane + 64 * k;
      if (v < V) out[v] += acc[tt][k];
    }
  }
}

// Same for d g: a wave owns UU label positions and loops over t.
template <int KE, int UU>
__device__ __forceinline__ void joint_fix_g(int block, const float* __restrict__ f, const float* __restrict__ g,
                                            float* __restrict__ dg, const int32_t* __restrict__ in_len,
                                            const int32_t* __restrict__ lab_len,
                                            const float* __restrict__ grad_scale, const JointGeom& jg, const Ws& w) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ntile = (jg.maxU1 + UU - 1) / UU;
  const int blocks_per_b = (ntile + 3) / 4;
  const int b = block / blocks_per_b;
  const int tile = (block - b * blocks_per_b) * 4 + wv;
  if (tile >= ntile) return;
  const int u0 = tile * UU;
  const int V = jg.V, maxU1 = jg.maxU1;
  int T, U;
  utt_dims(in_len, lab_len, b, jg.maxT, maxU1, T, U);
  const float sc = grad_scale ? grad_scale[b] : 1.f;
  const int nu = (T > 0) ? min(UU, max(U + 1 - u0, 0)) : 0;
  if (nu <= 0) return;
  float acc[UU][KE];
  bool any = false;
#pragma unroll
  for (int uu = 0; uu < UU; ++uu)
#pragma unroll
    for (int k = 0; k < KE; ++k) acc[uu][k] = 0.f;
  float g2[UU][KE];
#pragma unroll
  for (int uu = 0; uu < UU; ++uu) {
    const int u = min(u0 + uu, U);
    jrow_load<KE>(g + ((size_t)b * maxU1 + u) * V, V, lane, w.mg2[b * maxU1 + u], g2[uu]);
  }
  const float* fb = f + (size_t)b * jg.fstride;
  const float* mf = w.mf2 + b * jg.maxT;
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int uu = 0; uu < UU; ++uu) {
      if (uu >= nu) continue;
      const float4 cf = w.coef[((size_t)b * jg.maxT + t) * maxU1 + u0 + uu];  // uniform over the wave
      if (!(__float_as_int(cf.w) & kJointFlagBit)) continue;
      any = true;
      float f2[KE];
      jrow_load<KE>(fb + (size_t)t * V, V, lane, mf[t], f2);
#pragma unroll
      for (int k = 0; k < KE; ++k) acc[uu][k] += sc * fast_exp2(f2[k] + g2[uu][k] + cf.x);
    }
  }
  if (!any) return;
#pragma unroll
  for (int uu = 0; uu < UU; ++uu) {
    if (uu >= nu) continue;
    float* out = dg + ((size_t)b * maxU1 + u0 + uu) * V;
#pragma unroll
    for (int k = 0; k < KE; ++k) {
      const int v = lane + 64 * k;
      if (v < V) out[v] += acc[uu][k];
    }
  }
}

// Both exponent-domain passes in one launch (they touch different outputs): the first f_blocks blocks take d f,
// the rest d g.  With no flagged cell -- the usual case -- every block returns at once.
template <int KE, int TT>
__global__ __launch_bounds__(256) void k_joint_fix(const float* __restrict__ f, const float* __restrict__ g,
                                                   float* __restrict__ df, float* __restrict__ dg,
                                                   const int32_t* __restrict__ in_len,
                                                   const int32_t* __restrict__ lab_len,
                                                   const float* __restrict__ grad_scale, JointGeom jg, Ws w,
                                                   int f_blocks) {
  if (w.hdr->n_fallback == 0) return;
  if ((int)blockIdx.x < f_blocks) joint_fix_f<KE, TT>(blockIdx.x, f, g, df, in_len, lab_len, grad_scale, jg, w);
  else joint_fix_g<KE, TT>(blockIdx.x - f_blocks, f, g, dg, in_len, lab_len, grad_scale, jg, w);
}

}  // namespace
