// Part of the single translation unit rnnt_mi355x.hip: constants, the workspace header and layout, launch geometry of the row kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/rnnt_mi355x.h"

namespace {

constexpr float kLog2e = 1.4426950408889634f;
constexpr double kLn2 = 0.6931471805599453;
constexpr float kNegF = -1.0e30f;  // "log 0" that stays finite under addition
constexpr double kNeg = -1.0e30;
constexpr int kMaxU1 = 1024;
constexpr int kRowBlock = 256;  // threads per block of the streaming kernels
constexpr int kRowWaves = kRowBlock / 64;
// Lattice cells whose occupancy gamma = exp(alpha+beta-ll) is below 2^kPruneLog2 (3.7e-9) get an exactly zero
// gradient row: every element of that row is bounded by gamma, a sixteenth of the float32 spacing at 1 and more
// than four orders below the 1e-4 bar on gradients; measured at T=400, U=100, V=1024 the sums over u and over t
// that the caller forms downstream change by at most 2e-8 and 1e-7.  Their logits are then not read a second
// time.  This is the default; the threshold in force is a
// process-wide setting (rnnt_mi355x_set_prune_log2) handed to the gradient kernels as an argument.
constexpr float kPruneLog2 = -28.f;

// First bytes of the workspace, written on the device.
struct Header {
  int32_t flags;          // OR of RNNT_MI355X_FLAG_*
  int32_t n_partial;      // entries of `partial` the last k_grad wrote
  uint64_t partial_off;   // byte offset of `partial` from the workspace base
  int32_t n_fallback;     // fused joint: lattice cells whose linear-domain sum underflowed (handled by the exp path)
  int32_t pad0;
  uint64_t count_off;     // byte offset of `pcount` from the workspace base
  uint64_t n_cells;       // written by k_reduce_counts: lattice cells the last gradient pass visited ...
  uint64_t n_pruned;      // ... and how many of them it pruned
  int32_t n_count;        // entries of `pcount` the last gradient pass wrote
  int32_t n_items_ready;  // split gradient pass: work items of k_grad_rmw
  // ... while k_classify_rows runs: low half = work items handed out so far, high half = blocks that have taken
  // theirs (one atomic per block; zero between gradient passes)
  unsigned long long items_ctr;
  uint32_t rmw_next;      // work items of k_grad_rmw handed out beyond each block's first (zero between gradient passes)
  int32_t pad[47];
};
static_assert(sizeof(Header) == 256, "header is one 256-byte block");

struct Ws {
  Header* hdr;
  double* ll;      // [2][B]  log2-likelihood from alpha / from beta
  float* partial;  // per-block sum of squared gradient elements (ws_partial_slots entries)
  uint2* pcount;   // per-block {lattice cells, cells pruned} of the last gradient pass (same number of slots)
  // [B][maxT+2][maxU1] {log2 P(blank|t,u), log2 P(label u+1|t,u), log2-sum-exp2 of the row, 0}; frame t lives at
  // index t+1, frames -1 and maxT are guards.  Every entry that is not a lattice cell (guards, t >= T_b,
  // u > U_b) and the label slot of column U_b hold "log 0", so the sweeps need no masks.
  float4* rec;
  double* alpha;   // [B][D][U1s]
  double* beta;    // [B][D][U1s]
  // more than 128 label positions only: {log2 P(blank), log2 P(label)} of the cell (d - u, u) at [B][Dp][U1s], "log 0"
  // where there is no such cell -- the records by anti-diagonal, so that a wavefront of the wide sweep reads the 64
  // cells of a diagonal it owns as one contiguous run
  float2* lp2;
  int Dp;          // rows of lp2 per utterance = D rounded up to the sweep's band + one band
  // Split gradient pass (k_classify_rows / k_fill_rows / k_grad_rmw, see grad_split.h).  An utterance's rows
  // r = t*maxU1 + u are cut into regions of kRegionRows consecutive rows:
  float4* crec;       // [B][regions][kRegionStride] {c2, pb, pl, (row in region) << 16 | label} (cell_coef.h): from the
                      // front the region's full rows in row order, then entries that stand for nothing up to a whole
                      // work item; from the end backwards its sparse rows
  int* rcount;        // [B][regions] sparse rows of the region
  int* items;         // work items of k_grad_rmw: (region id << 8) | block of entries inside the region's entries
  uint32_t* rowbits;  // [B][row_words] bit r of an utterance set: row r is full (k_grad_rmw writes it, not k_fill_rows)
  int row_words;      // 32-bit words of rowbits per utterance (rows rounded up to a multiple of 64)
  // fused additive joint only:
  float* mf2;      // [B][maxT]  max_v f(b,t,v) * log2(e)
  float* mg2;      // [B][maxU1] max_v g(b,u,v) * log2(e)
  float* fblank;   // [B][maxT]  f(b,t,blank)*log2(e) - mf2
  float* gblank;   // [B][maxU1] g(b,u,blank)*log2(e) - mg2
  float* glabel;   // [B][maxU1] g(b,u,y_{u+1})*log2(e) - mg2 (label clamped into [0,V); unused at u >= U_b)
  float* flab;     // [B][maxT][maxU1] f(b,t,y_{u+1})*log2(e) - mf2 of the lattice cells.  Lives in the storage of Ctu:
                   // the forward is done with it before a backward writes Ctu
  float4* coef;    // [B][maxT][maxU1] {c, pb, pl, label code} per cell (c = -inf: no gradient)
  // Operands of the three contractions, zero-padded, each stored in tiles of 32 rows x 8 columns
  // ([row tile][column chunk][32][8], see tiled_off) so that every wavefront load is a run of whole cache lines:
  float* Fp;       // [B] rows t < Tp,  columns v < Vp:  2^(f*log2e - mf2)
  float* Gp;       // [B] rows u < U1p, columns v < Vp:  2^(g*log2e - mg2)
  float* Ctu;      // [B] rows t < Tp,  columns u < U1p: C = s_b * gamma / (cell sum); 0 where nothing flows through the products
  float* Cut;      // [B] rows u < U1p, columns t < Tp:  the same numbers transposed
  int* head_of;    // [B][Vp]       first position u whose target symbol is v (the blank's is U_b + 1), or -1
  int* nxt;        // [B][maxU1+2]  next position with the same target as position u, or -1
  float* rowpart;  // [B][Tp][U1p/32] sums of s*pb(t,u) over runs of 32 positions
  float2* colpart; // [B][Tp/8][U1p] {sum of s*pb, sum of s*pl} over runs of 8 frames
  float* dgpart;   // [dg_split][B][maxU1][V] partial d g of each slice of the frames (only when dg_split > 1)
  int dg_split;    // slices the sum over frames of the d g contraction is cut into (1 = written straight to d g)
  int Vp, U1p, Tp; // padded extents: maxT and maxU1 rounded up to 32, V to 32 / 64 / a multiple of 128
  size_t partial_off, count_off;
  int D;    // diagonals = maxT + maxU1 - 1
  int U1s;  // row stride = 64 * sweep_cols_per_lane(maxU1)
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Lattice columns each lane of a sweep wavefront owns (1, 2, 4, 8 or 16).  The
// row stride of the per-cell arrays is 64 times this, so that every lane of a
// sweep can load and store its columns unconditionally.
inline int sweep_cols_per_lane(int maxU1) {
  return maxU1 <= 64 ? 1 : maxU1 <= 128 ? 2 : maxU1 <= 256 ? 4 : maxU1 <= 512 ? 8 : 16;
}

// Above 128 label positions the sweeps run one wavefront per 64 lattice columns (k_sweep_wide) over the records
// re-ordered by anti-diagonal (k_skew_records); up to 128 one wavefront per utterance and direction (k_sweep).
#ifndef RNNT_SWEEP_BAND
#define RNNT_SWEEP_BAND 32
#endif
#ifndef RNNT_SWEEP_WIDE_ABOVE  // (64 measured: 83 us against 63 us at T=500, U+1=101 -- the re-ordering pass and the band of skew cost more than one column per lane gains)
#define RNNT_SWEEP_WIDE_ABOVE 128
#endif
constexpr int kSweepBand = RNNT_SWEEP_BAND;  // diagonals between two block barriers of the wide sweep
inline bool sweep_is_wide(int maxU1) {
#ifdef RNNT_NO_WIDE_SWEEP
  return false;
#else
  return maxU1 > RNNT_SWEEP_WIDE_ABOVE;
#endif
}

// The d g contraction has only B * (U1p/32) * (column strips) wavefront tiles, each summing over all frames: for a
// small vocabulary and few utterances that is a fraction of the 1024 SIMDs with a long serial loop.  The sum over
// frames is then cut into 2, 4 or 8 slices whose partial results k_joint_corr_g adds up in a fixed order.
inline int joint_dg_split(int B, int Tp, int U1p, int Vp) {
  const int nt = Vp <= 32 ? 1 : Vp <= 64 ? 2 : 4;
  const long long waves = (long long)B * (U1p / 32) * (Vp / (32 * nt));
  int ks = 1;
  while (waves * ks < 768 && ks < 8 && Tp / (ks * 2) >= 64) ks *= 2;
  return ks;
}

// Slots of the per-block arrays `partial` / `pcount`: one per block of the gradient pass, whose blocks own
// TT >= 1 frames (k_grad), a share of the rows (k_grad_small: fewer blocks than B*maxT), at least kMinSpanRows
// consecutive rows of an utterance (k_grad_span) or at least kMinRmwRows entries of a region of kRegionRows rows
// (k_grad_rmw).
constexpr int kMinSpanRows = 8;
#ifndef RNNT_REGION_ROWS
#define RNNT_REGION_ROWS 2048
#endif
constexpr int kRegionRows = RNNT_REGION_ROWS;
constexpr int kRegionStride = kRegionRows + 64;  // entries of crec per region: its rows, padded up to a whole work item
constexpr int kMinRmwRows = 8;
inline int ws_regions(int maxT, int maxU1) { return (int)(((size_t)maxT * maxU1 + kRegionRows - 1) / kRegionRows); }
inline size_t ws_partial_slots(int B, int maxT, int maxU1) {
  const size_t span_blocks = ((size_t)maxT * maxU1 + kMinSpanRows - 1) / kMinSpanRows;
  const size_t rmw_blocks = (size_t)ws_regions(maxT, maxU1) * ((kRegionRows + kMinRmwRows - 1) / kMinRmwRows + 1);
  size_t per_utt = span_blocks > (size_t)maxT ? span_blocks : (size_t)maxT;
  if (rmw_blocks > per_utt) per_utt = rmw_blocks;
  return (size_t)B * per_utt;
}

// Lays the workspace out; returns total bytes.  base may be null (size query).
size_t ws_layout(void* base, int B, int maxT, int maxU1, Ws* w, bool joint = false, int V = 0) {
  const int D = maxT + maxU1 - 1;
  const int U1s = 64 * sweep_cols_per_lane(maxU1);
  const size_t plane = (size_t)B * D * U1s;
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  const size_t o_hdr = take(sizeof(Header));
  const size_t o_ll = take(sizeof(double) * 2 * B);
  const size_t n_part = ws_partial_slots(B, maxT, maxU1);
  const size_t o_partial = take(sizeof(float) * n_part);
  const size_t o_pcount = take(sizeof(uint2) * n_part);
  const size_t o_rec = take(sizeof(float4) * (size_t)B * (maxT + 2) * maxU1);
  const size_t o_alpha = take(sizeof(double) * plane);
  const size_t o_beta = take(sizeof(double) * plane);
  const int Dp = sweep_is_wide(maxU1) ? (D + kSweepBand - 1) / kSweepBand * kSweepBand + kSweepBand : 0;
  const size_t o_lp2 = Dp ? take(sizeof(float2) * (size_t)B * Dp * U1s) : 0;
  const int row_words = (int)(((size_t)maxT * maxU1 + 63) / 64 * 2);
  const size_t n_regions = (size_t)B * ws_regions(maxT, maxU1);
  const size_t o_crec = joint ? 0 : take(sizeof(float4) * n_regions * kRegionStride);
  const size_t o_rcount = joint ? 0 : take(sizeof(int) * n_regions);
  const size_t o_items = joint ? 0 : take(sizeof(int) * n_regions * ((kRegionRows + kMinRmwRows - 1) / kMinRmwRows));
  const size_t o_rowbits = joint ? 0 : take(sizeof(uint32_t) * (size_t)B * row_words);
  const size_t o_mf2 = joint ? take(sizeof(float) * (size_t)B * maxT) : 0;
  const size_t o_mg2 = joint ? take(sizeof(float) * (size_t)B * maxU1) : 0;
  const size_t o_fblank = joint ? take(sizeof(float) * (size_t)B * maxT) : 0;
  const size_t o_gblank = joint ? take(sizeof(float) * (size_t)B * maxU1) : 0;
  const size_t o_glabel = joint ? take(sizeof(float) * (size_t)B * maxU1) : 0;
  const size_t o_coef = joint ? take(sizeof(float4) * (size_t)B * maxT * maxU1) : 0;
  const int Vp = V <= 32 ? 32 : V <= 64 ? 64 : (int)align_up((size_t)V, 128);
  const int U1p = (int)align_up((size_t)maxU1, 32), Tp = (int)align_up((size_t)maxT, 32);
  const size_t o_Fp = joint ? take(sizeof(float) * (size_t)B * Tp * Vp) : 0;
  const size_t o_Gp = joint ? take(sizeof(float) * (size_t)B * U1p * Vp) : 0;
  const size_t o_Ctu = joint ? take(sizeof(float) * (size_t)B * Tp * U1p) : 0;
  const size_t o_Cut = joint ? take(sizeof(float) * (size_t)B * Tp * U1p) : 0;
  const size_t o_head = joint ? take(sizeof(int) * (size_t)B * Vp) : 0;
  const size_t o_nxt = joint ? take(sizeof(int) * (size_t)B * (maxU1 + 2)) : 0;
  const size_t o_rowpart = joint ? take(sizeof(float) * (size_t)B * Tp * (U1p / 32)) : 0;
  const size_t o_colpart = joint ? take(sizeof(float2) * (size_t)B * (Tp / 8) * U1p) : 0;
  const int dg_split = joint ? joint_dg_split(B, Tp, U1p, Vp) : 1;
  const size_t o_dgpart = dg_split > 1 ? take(sizeof(float) * (size_t)dg_split * B * maxU1 * V) : 0;
  if (w) {
    w->hdr = reinterpret_cast<Header*>(p + o_hdr);
    w->ll = reinterpret_cast<double*>(p + o_ll);
    w->partial = reinterpret_cast<float*>(p + o_partial);
    w->pcount = reinterpret_cast<uint2*>(p + o_pcount);
    w->count_off = o_pcount;
    w->rec = reinterpret_cast<float4*>(p + o_rec);
    w->alpha = reinterpret_cast<double*>(p + o_alpha);
    w->beta = reinterpret_cast<double*>(p + o_beta);
    w->lp2 = Dp ? reinterpret_cast<float2*>(p + o_lp2) : nullptr;
    w->Dp = Dp;
    w->crec = joint ? nullptr : reinterpret_cast<float4*>(p + o_crec);
    w->rcount = joint ? nullptr : reinterpret_cast<int*>(p + o_rcount);
    w->items = joint ? nullptr : reinterpret_cast<int*>(p + o_items);
    w->rowbits = joint ? nullptr : reinterpret_cast<uint32_t*>(p + o_rowbits);
    w->row_words = row_words;
    w->mf2 = joint ? reinterpret_cast<float*>(p + o_mf2) : nullptr;
    w->mg2 = joint ? reinterpret_cast<float*>(p + o_mg2) : nullptr;
    w->fblank = joint ? reinterpret_cast<float*>(p + o_fblank) : nullptr;
    w->gblank = joint ? reinterpret_cast<float*>(p + o_gblank) : nullptr;
    w->glabel = joint ? reinterpret_cast<float*>(p + o_glabel) : nullptr;
    w->coef = joint ? reinterpret_cast<float4*>(p + o_coef) : nullptr;
    w->Fp = joint ? reinterpret_cast<float*>(p + o_Fp) : nullptr;
    w->Gp = joint ? reinterpret_cast<float*>(p + o_Gp) : nullptr;
    w->Ctu = joint ? reinterpret_cast<float*>(p + o_Ctu) : nullptr;
    w->flab = w->Ctu;  // B*maxT*maxU1 <= B*Tp*U1p floats
    w->Cut = joint ? reinterpret_cast<float*>(p + o_Cut) : nullptr;
    w->head_of = joint ? reinterpret_cast<int*>(p + o_head) : nullptr;
    w->nxt = joint ? reinterpret_cast<int*>(p + o_nxt) : nullptr;
    w->rowpart = joint ? reinterpret_cast<float*>(p + o_rowpart) : nullptr;
    w->colpart = joint ? reinterpret_cast<float2*>(p + o_colpart) : nullptr;
    w->dgpart = dg_split > 1 ? reinterpret_cast<float*>(p + o_dgpart) : nullptr;
    w->dg_split = dg_split;
    w->Vp = Vp; w->U1p = U1p; w->Tp = Tp;
    w->partial_off = o_partial;
    w->D = D;
    w->U1s = U1s;
  }
  return off;
}

// Geometry of the streaming kernels: one block per (utterance, TT frames).
struct RowGeom {
  int maxT, maxU1, V, blank;
  int TT;     // frames per block
  int nTB;    // frame blocks per utterance
  int npass;  // passes of NC chunks per lane over one row
  int b0;     // first utterance of this launch
  float prune_log2;  // cells with log2 gamma below this get a zero gradient row without being re-read
  size_t xstride;   // elements between consecutive utterances of the logits (>= maxT*maxU1*V)
  size_t gstride;   // same for the gradient buffer
  int nblocks;      // blocks of this launch = (utterances of this launch) * nTB
  int nblocks_all;  // blocks of the whole batch = B * nTB
};

}  // namespace
