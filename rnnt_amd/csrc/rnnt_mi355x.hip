// RNN-Transducer loss for MI355X (gfx950): three hand-written kernels behind the
// C ABI declared in include/rnnt_mi355x.h.
//
//   k_lse   : one streaming pass over logits[B,T,U1,V]; per lattice cell the
//             log-sum-exp over V plus the blank / next-label log-probabilities
//             (the log_softmax of reference rnnt/base/util.py:66,109 reduced to
//             the three numbers per cell the lattice needs).
//   k_sweep : alpha and beta recursions of the transducer lattice (what
//             rnnt_loss computes, call sites reference util.py:70-72,113-115) as
//             anti-diagonal sweeps, one wavefront per (utterance, direction).
//   k_grad  : second streaming pass; writes d cost/d logits (log-softmax
//             backward folded in, upstream scale of reference
//             rnnt/keras_model.py:215-217 folded in) over the logits.
//
// All lattice quantities are kept in base-2 logarithms (v_exp_f32 / v_log_f32
// are base 2).  alpha/beta are accumulated in float64 with the small
// log2(1+2^-d) correction evaluated in float32: a pure-float32 lattice drifts
// by ~1e-3 in the gradient at T=400,U=100 because alpha and beta reach
// magnitudes of several thousand.
//
// Per-cell arrays in the workspace use a skewed layout [b][d = t+u][u] so that
// every anti-diagonal is one contiguous row: the sweeps read and write
// coalesced rows and never touch LDS banks with a stride.
//
// One translation unit.  This file holds the host side (shape checks, kernel
// selection, launches) and the C entry points; the kernels are in the pieces
// included below:
//   workspace.h      constants, workspace header and layout
//   device_util.h    exp2/log2, vector loads, DPP reductions, row cursor
//   cell_coef.h      a cell's gradient coefficients {c2, pb, pl, label}; full / sparse / pruned rows
//   lse_span.h       k_lse_span (aligned rows <= 4096)
//   lse_rows.h       k_lse                 grad_rows.h   k_grad (any row)   grad_span.h  k_grad_span (aligned rows <= 4096)
//   sweep.h          k_sweep               small_vocab.h V <= 128 variants of both passes
//   joint.h          additive joint fused into the loss (matrix-core contractions)
//   aux_kernels.h    zero-fill of cropped frames, workspace init, batch scalars
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/rnnt_mi355x.h"

#include "workspace.h"
#include "device_util.h"
#include "cell_coef.h"
#include "lse_rows.h"
#include "sweep.h"
#include "grad_rows.h"
#include "grad_span.h"
#include "grad_split.h"
#include "grad_flat.h"
#include "lse_span.h"
#include "small_vocab.h"
#include "joint.h"
#include "aux_kernels.h"

#include <atomic>
#include <mutex>

namespace {

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

// Default pruning threshold of the gradient kernels (cell_coef.h) for calls that pass NaN as theirs: the library's one
// process-wide setting, see rnnt_mi355x_set_prune_log2.
std::atomic<float> g_prune_log2{kPruneLog2};

// The threshold a call runs with: its own argument, or the process default when that is NaN.
int resolve_prune(float arg, float* out) {
  if (arg != arg) {
    *out = g_prune_log2.load(std::memory_order_relaxed);
    return RNNT_MI355X_OK;
  }
  if (!(arg <= -16.f)) return RNNT_MI355X_INVALID_VALUE;
  *out = arg;
  return RNNT_MI355X_OK;
}

struct RowCfg {
  int vec, L, nc, g;
  RowGeom gm;
};

RowCfg pick_row_cfg(int B, int maxT, int maxU1, int V, int blank, const void* p0, const void* p1) {
  RowCfg c;
  const uintptr_t a = reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1);
  c.vec = (V % 4 == 0 && a % 16 == 0) ? 4 : (V % 2 == 0 && a % 8 == 0) ? 2 : 1;
  const int nchunk = V / c.vec;
  // lanes per row: one vector load of the group covers as much of the row as possible (whole-row contiguous
  // accesses; partial-line non-temporal stores are slow), between 8 and 64 lanes
  c.L = 8;
  while (c.L < 64 && c.L < nchunk) c.L <<= 1;
  // Rows of 17..64 chunks: 16 lanes per row with 2 or 4 chunks each instead of 32 / 64 lanes with one; rows of
  // 65..128 chunks: 32 lanes with 4 each instead of 64 with 2.  The per-row cost (two reductions over the group,
  // the record, the cursor) is then paid once per 2-4 loads, the reductions stay inside one or two DPP rows and
  // a wavefront works on several rows at a time.  Measured at B=64, T=500, U=100 (first / second pass, TB/s):
  // V=160 2.3 / 1.8 -> 4.7 / 7.0; V=256 3.5 / 2.7 -> 5.0 / 7.0; V=384 4.6 / 5.5 -> 5.4 / 7.2; V=512 5.1 / 6.1 ->
  // 5.5 / 7.1.
  if (nchunk > 16 && nchunk <= 64) c.L = 16;
  else if (nchunk > 64 && nchunk <= 128) c.L = 32;
  const int cpl = (nchunk + c.L - 1) / c.L;  // chunks per lane per row
  c.nc = cpl >= 3 ? 4 : cpl;
  c.g = 4 / c.nc;
  const int rows_per_iter = kRowWaves * c.g * (64 / c.L);
  // frames per block: about 24 tiles per wave, so that the pipeline fill/drain is amortised
#ifndef RNNT_ROW_TILES
#define RNNT_ROW_TILES 24
#endif
  int TT = (RNNT_ROW_TILES * rows_per_iter + maxU1 - 1) / maxU1;
  while (TT > 1 && TT * maxU1 > 2048) --TT;  // the block's records are staged in LDS (<= 32 KiB)
  if (TT < 1) TT = 1;
  if (TT > maxT) TT = maxT;
  c.gm.maxT = maxT;
  c.gm.maxU1 = maxU1;
  c.gm.V = V;
  c.gm.blank = blank;
  c.gm.TT = TT;
  c.gm.nTB = (maxT + TT - 1) / TT;
  c.gm.npass = (cpl + c.nc - 1) / c.nc;
  c.gm.b0 = 0;
  c.gm.prune_log2 = kPruneLog2;  // (the caller sets the threshold of its call)
  c.gm.xstride = c.gm.gstride = (size_t)maxT * maxU1 * V;
  c.gm.nblocks = B * c.gm.nTB;
  c.gm.nblocks_all = B * c.gm.nTB;
  return c;
}

#define RNNT_DISPATCH_LNG(KERNEL, VEC, cfg, LDS, ...)                                                     \
  do {                                                                                                    \
    const dim3 grid_((cfg).gm.nblocks), blk_(kRowBlock);                                                  \
    if ((cfg).L == 64 && (cfg).nc == 4) KERNEL<VEC, 64, 4, 1><<<grid_, blk_, LDS, st>>>(__VA_ARGS__);      \
    else if ((cfg).L == 64 && (cfg).nc == 2) KERNEL<VEC, 64, 2, 2><<<grid_, blk_, LDS, st>>>(__VA_ARGS__); \
    else if ((cfg).L == 64) KERNEL<VEC, 64, 1, 4><<<grid_, blk_, LDS, st>>>(__VA_ARGS__);                  \
    else if ((cfg).L == 32 && (cfg).nc == 4) KERNEL<VEC, 32, 4, 1><<<grid_, blk_, LDS, st>>>(__VA_ARGS__); \
    else if ((cfg).L == 32 && (cfg).nc == 2) KERNEL<VEC, 32, 2, 2><<<grid_, blk_, LDS, st>>>(__VA_ARGS__); \
    else if ((cfg).L == 32) KERNEL<VEC, 32, 1, 4><<<grid_, blk_, LDS, st>>>(__VA_ARGS__);                  \
    else if ((cfg).L == 16 && (cfg).nc == 4) KERNEL<VEC, 16, 4, 1><<<grid_, blk_, LDS, st>>>(__VA_ARGS__); \
    else if ((cfg).L == 16 && (cfg).nc == 2) KERNEL<VEC, 16, 2, 2><<<grid_, blk_, LDS, st>>>(__VA_ARGS__); \
    else if ((cfg).L == 16) KERNEL<VEC, 16, 1, 4><<<grid_, blk_, LDS, st>>>(__VA_ARGS__);                  \
    else KERNEL<VEC, 8, 1, 4><<<grid_, blk_, LDS, st>>>(__VA_ARGS__);                                      \
  } while (0)

#define RNNT_DISPATCH(KERNEL, cfg, LDS, ...)                                      \
  do {                                                                            \
    if ((cfg).vec == 4) RNNT_DISPATCH_LNG(KERNEL, 4, cfg, LDS, __VA_ARGS__);      \
    else if ((cfg).vec == 2) RNNT_DISPATCH_LNG(KERNEL, 2, cfg, LDS, __VA_ARGS__); \
    else RNNT_DISPATCH_LNG(KERNEL, 1, cfg, LDS, __VA_ARGS__);                     \
  } while (0)

int check_shape(int B, int maxT, int maxU1, int V, int blank) {
  if (B < 1 || maxT < 1 || maxU1 < 1 || V < 1) return RNNT_MI355X_INVALID_VALUE;
  if (blank < 0 || blank >= V) return RNNT_MI355X_INVALID_VALUE;
  if (maxU1 > kMaxU1) return RNNT_MI355X_UNSUPPORTED_SIZE;
  if ((long long)B * maxT * maxU1 >= (1LL << 31)) return RNNT_MI355X_UNSUPPORTED_SIZE;
  // the sweeps address an utterance's records with signed 32-bit byte offsets (16 bytes a record; their cursors run
  // up to maxU1 + 32 frames past either end before they are clamped) and 24-bit frame indices
  if (((long long)maxT + maxU1 + 32) * maxU1 >= (1LL << 27) || maxT >= (1 << 24) - 2) return RNNT_MI355X_UNSUPPORTED_SIZE;
  return RNNT_MI355X_OK;
}

// compute units of the current device (a resident grid is sized from it)
int compute_units() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
    n = 256;
  return n;
}

int launch_status() { return hipGetLastError() == hipSuccess ? RNNT_MI355X_OK : RNNT_MI355X_LAUNCH_FAILURE; }

struct Call {
  const float* logits;
  float* grad;
  const int32_t* labels;
  const int32_t* in_len;
  const int32_t* lab_len;
  int B, maxT, maxU1, V, blank;
  size_t xstride, gstride;  // elements between utterances in logits / grad
  int t_off = 0;            // crop: the lattice starts at frame t_off of [B, t_off + maxT, maxU1, V] tensors whose
                            // frame 0 is logits - t_off*maxU1*V (and grad likewise); 0 = no such promise
  const float* grad_scale;
  float* costs;
  float prune_log2 = kPruneLog2;  // threshold of this call's gradient pass
  Ws w;
};

RowCfg sub_cfg(const Call& c, const void* p1, int b0, int nb) {
  // the widest vector access must also survive the batch strides
  const uintptr_t align_bits = reinterpret_cast<uintptr_t>(p1) | (uintptr_t)(c.xstride * 4) | (uintptr_t)(c.gstride * 4);
  RowCfg cfg = pick_row_cfg(c.B, c.maxT, c.maxU1, c.V, c.blank, c.logits, reinterpret_cast<const void*>(align_bits));
  cfg.gm.xstride = c.xstride;
  cfg.gm.gstride = c.gstride;
  cfg.gm.prune_log2 = c.prune_log2;
  cfg.gm.b0 = b0;
  cfg.gm.nblocks = nb * cfg.gm.nTB;
  return cfg;
}

// Small-vocabulary path: V <= 64 and 16-byte aligned buffers (a tile of 64 rows is then a whole number of float4).
// The small-vocabulary kernels tile the tensor in runs of 64 rows across utterance boundaries: it has to be
// dense, possibly with t_off leading frames per utterance that are not part of the lattice.
bool use_small(const Call& c, const void* p1) {
  const size_t frame = (size_t)c.maxU1 * c.V;
  const uintptr_t a = reinterpret_cast<uintptr_t>(c.logits - c.t_off * frame) |
                      (p1 ? reinterpret_cast<uintptr_t>(static_cast<const float*>(p1) - c.t_off * frame) : 0);
  const size_t dense = (size_t)(c.t_off + c.maxT) * frame;
  return c.V <= kSmallMaxV && a % 16 == 0 && c.xstride == dense && c.gstride == dense;
}

SmallGeom small_geom(const Call& c, int* blocks) {
  SmallGeom g;
  g.B = c.B; g.maxT = c.maxT; g.maxU1 = c.maxU1; g.V = c.V; g.blank = c.blank;
  g.frames = c.t_off + c.maxT; g.t_off = c.t_off;
  g.total_rows = (long long)c.B * g.frames * c.maxU1;
  g.prune_log2 = c.prune_log2;
  g.ntiles_full = (int)(g.total_rows / 64);
  long long want = (g.ntiles_full + kSmallWaves - 1) / kSmallWaves;
  // persistent grid, exactly one resident round: tiles are dealt round-robin to the waves.  Residency per CU is
  // limited by LDS (160 KiB / tile bytes per block) and by registers (6 blocks of two waves).
  const long long lds_bytes = 16LL * small_lds_f4(c.V, true) * kSmallWaves;  // the larger of the two passes' tiles
  long long per_cu = (160LL * 1024) / lds_bytes;
  if (per_cu > 6) per_cu = 6;
  if (per_cu < 1) per_cu = 1;
  const long long cap = 256 * per_cu;
  const long long cap2 = (long long)c.B * c.maxT;      // one gradient-norm partial per block
  if (want > cap) want = cap;
  if (want > cap2) want = cap2;
  if (want < 1) want = 1;
  *blocks = (int)want;
  return g;
}

// Instantiation of a small-vocabulary kernel for V: vector width of the row loops, LDS padding mode, staging depth.
// Tiles above 64 KiB per block need the dynamic-LDS limit raised.
#define RNNT_SMALL_LAUNCH(KERNEL, VEC, PAD, NLD, blocks, lds, st, ...)                                          \
  do {                                                                                                          \
    if ((lds) > 48 * 1024) /* per device, so not cached in a process-wide flag */                               \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL<VEC, PAD, NLD>),                          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);                         \
    KERNEL<VEC, PAD, NLD><<<blocks, kSmallBlock, lds, st>>>(__VA_ARGS__);                                       \
  } while (0)
#define RNNT_SMALL_DISPATCH(KERNEL, V, GRAD, blocks, lds, st, ...)                                              \
  do {                                                                                                          \
    const int pm = small_pad_mode(V, GRAD);                                                                     \
    if ((V) <= 64) {                                                                                            \
      if (pm == 1) RNNT_SMALL_LAUNCH(KERNEL, 4, 1, 16, blocks, lds, st, __VA_ARGS__);                           \
      else if ((V) % 4 == 0) RNNT_SMALL_LAUNCH(KERNEL, 4, 0, 16, blocks, lds, st, __VA_ARGS__);                 \
      else if ((V) % 2 == 0) RNNT_SMALL_LAUNCH(KERNEL, 2, 0, 16, blocks, lds, st, __VA_ARGS__);                 \
      else RNNT_SMALL_LAUNCH(KERNEL, 1, 0, 16, blocks, lds, st, __VA_ARGS__);                                   \
    } else {                                                                                                    \
      if (pm == 2) RNNT_SMALL_LAUNCH(KERNEL, 4, 2, 32, blocks, lds, st, __VA_ARGS__);                           \
      else if ((V) % 4 == 0) RNNT_SMALL_LAUNCH(KERNEL, 4, 0, 32, blocks, lds, st, __VA_ARGS__);                 \
      else if ((V) % 2 == 0) RNNT_SMALL_LAUNCH(KERNEL, 2, 0, 32, blocks, lds, st, __VA_ARGS__);                 \
      else RNNT_SMALL_LAUNCH(KERNEL, 1, 0, 32, blocks, lds, st, __VA_ARGS__);                                   \
    }                                                                                                           \
  } while (0)

// the same instantiations for the one-shot form of the gradient pass (one wavefront per block)
#define RNNT_SMALL_SPAN_LAUNCH(KERNEL, VEC, PAD, NLD, blocks, lds, st, ...)                                     \
  do {                                                                                                          \
    if ((lds) > 48 * 1024) /* per device, so not cached in a process-wide flag */                               \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL<VEC, PAD, NLD>),                          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);                         \
    KERNEL<VEC, PAD, NLD><<<blocks, 64, lds, st>>>(__VA_ARGS__);                                                \
  } while (0)
#define RNNT_SMALL_SPAN_DISPATCH(KERNEL, V, GRAD, blocks, lds, st, ...)                                         \
  do {                                                                                                          \
    const int pm = small_pad_mode(V, GRAD);                                                                     \
    if ((V) <= 64) {                                                                                            \
      if (pm == 1) RNNT_SMALL_SPAN_LAUNCH(KERNEL, 4, 1, 16, blocks, lds, st, __VA_ARGS__);                      \
      else if ((V) % 4 == 0) RNNT_SMALL_SPAN_LAUNCH(KERNEL, 4, 0, 16, blocks, lds, st, __VA_ARGS__);            \
      else if ((V) % 2 == 0) RNNT_SMALL_SPAN_LAUNCH(KERNEL, 2, 0, 16, blocks, lds, st, __VA_ARGS__);            \
      else RNNT_SMALL_SPAN_LAUNCH(KERNEL, 1, 0, 16, blocks, lds, st, __VA_ARGS__);                              \
    } else {                                                                                                    \
      if (pm == 2) RNNT_SMALL_SPAN_LAUNCH(KERNEL, 4, 2, 32, blocks, lds, st, __VA_ARGS__);                      \
      else if ((V) % 4 == 0) RNNT_SMALL_SPAN_LAUNCH(KERNEL, 4, 0, 32, blocks, lds, st, __VA_ARGS__);            \
      else if ((V) % 2 == 0) RNNT_SMALL_SPAN_LAUNCH(KERNEL, 2, 0, 32, blocks, lds, st, __VA_ARGS__);            \
      else RNNT_SMALL_SPAN_LAUNCH(KERNEL, 1, 0, 32, blocks, lds, st, __VA_ARGS__);                              \
    }                                                                                                           \
  } while (0)

SpanGeom span_geom(const Call& c, int b0) {
  SpanGeom sg;
  sg.maxT = c.maxT; sg.maxU1 = c.maxU1; sg.V = c.V; sg.blank = c.blank;
  sg.rows_utt = c.maxT * c.maxU1;
  sg.nchunks = sg.nblocks = sg.nblocks_all = 0;  // set per kernel: they depend on its rows per block
  sg.b0 = b0;
  sg.prune_log2 = c.prune_log2;
  sg.xstride = c.xstride; sg.gstride = c.gstride;
  return sg;
}

void launch_lse(const Call& c, int b0, int nb, hipStream_t st) {
  if (b0 == 0 && nb == c.B && use_small(c, c.logits)) {
    int blocks;
    const SmallGeom g = small_geom(c, &blocks);
    const size_t lds = sizeof(float4) * (size_t)small_lds_f4(c.V, false) * kSmallWaves;
    const float* x0 = c.logits - (size_t)c.t_off * c.maxU1 * c.V;  // frame 0 of the uncropped tensor
#ifndef RNNT_NO_SMALL_LSE_SPAN
    // One-shot blocks up to 112 logits (B=64, T=500, U=100: V=32 95 -> 83 us, 62 159 -> 151, 80 291 -> 206, 100 309 -> 257;
    // 108..124 alike); the largest tiles keep the persistent, pipelined kernel (V=126 353 against 390 us, 128 315
    // against 357: with 32 KiB of LDS per tile too few one-shot wavefronts are resident to overlap on their own).
    if (c.V <= 112) {  // one block per tile (plus one for a partial last tile)
      const long long sblocks = (long long)g.ntiles_full + ((long long)g.ntiles_full * 64 < g.total_rows ? 1 : 0);
      const size_t slds = sizeof(float4) * (size_t)small_lds_f4(c.V, false);
      RNNT_SMALL_SPAN_DISPATCH(k_lse_small_span, c.V, false, (int)sblocks, slds, st, x0, c.labels, c.in_len, c.lab_len, g, c.w);
      return;
    }
#endif
    RNNT_SMALL_DISPATCH(k_lse_small, c.V, false, blocks, lds, st, x0, c.labels, c.in_len, c.lab_len, g, c.w);
    return;
  }
  const RowCfg cfg = sub_cfg(c, c.logits, b0, nb);
#ifndef RNNT_NO_LSE_SPAN
  // (measured: rows of 4 KiB and more gain 2-5 % over k_lse; 2 KiB rows of a ragged batch lose 10 %)
  if (cfg.vec == 4 && c.V > 512 && c.V <= 4096) {
    SpanGeom sg = span_geom(c, b0);
    const int nchunk = c.V / 4;
    auto go = [&](auto kernel, int rows_per_block) {
      sg.nchunks = (sg.rows_utt + rows_per_block - 1) / rows_per_block;
      sg.nblocks = nb * sg.nchunks;
      sg.nblocks_all = c.B * sg.nchunks;
      kernel<<<dim3((unsigned)sg.nblocks), dim3(kLseWaves * 64), 0, st>>>(c.logits, c.labels, c.in_len, c.lab_len, sg, c.w);
    };
    if (nchunk <= 256) go(k_lse_span<64, 4, RNNT_LSE_NG>, kLseWaves * RNNT_LSE_NG);
    else if (nchunk <= 512) go(k_lse_span<64, 8, RNNT_LSE_NG / 2>, kLseWaves * (RNNT_LSE_NG / 2));
    else go(k_lse_span<64, 16, RNNT_LSE_NG / 4>, kLseWaves * (RNNT_LSE_NG / 4));
    return;
  }
#endif
  RNNT_DISPATCH(k_lse, cfg, sizeof(float4) * (size_t)cfg.gm.TT * c.maxU1, c.logits, c.labels, c.in_len, c.lab_len,
                cfg.gm, c.w);
}

void launch_sweep(const Call& c, int b0, int nb, hipStream_t st) {
  const int grid = 2 * nb;
  if (sweep_is_wide(c.maxU1)) {  // (whole batch only: the re-ordered records are built for all of it)
    const int tpb = c.w.U1s < 256 ? c.w.U1s : 256;
    k_skew_records<<<dim3((unsigned)(c.w.U1s / tpb), (unsigned)(c.w.Dp / kSkewRows), (unsigned)c.B), tpb, 0, st>>>(
        c.w, c.maxT, c.maxU1);
    k_sweep_wide<<<grid, 64 * ((c.maxU1 + 63) / 64), 0, st>>>(c.w, c.costs, c.in_len, c.lab_len, c.B, c.maxT, c.maxU1, b0);
    return;
  }
  switch (sweep_cols_per_lane(c.maxU1)) {
    case 1: k_sweep<1><<<grid, 64, 0, st>>>(c.w, c.costs, c.in_len, c.lab_len, c.B, c.maxT, c.maxU1, b0); break;
    case 2: k_sweep<2><<<grid, 64, 0, st>>>(c.w, c.costs, c.in_len, c.lab_len, c.B, c.maxT, c.maxU1, b0); break;
    case 4: k_sweep<4><<<grid, 64, 0, st>>>(c.w, c.costs, c.in_len, c.lab_len, c.B, c.maxT, c.maxU1, b0); break;
    case 8: k_sweep<8><<<grid, 64, 0, st>>>(c.w, c.costs, c.in_len, c.lab_len, c.B, c.maxT, c.maxU1, b0); break;
    default: k_sweep<16><<<grid, 64, 0, st>>>(c.w, c.costs, c.in_len, c.lab_len, c.B, c.maxT, c.maxU1, b0); break;
  }
}

// Which kernel takes the gradient pass (measured at B=64, T=500, U=100: bytes read + written per second, flattered
// alike by pruned rows):
//   k_grad_flat<true>   16-byte rows that are no multiple of 512 bytes, and all of them up to 128 logits: 5.9-8.4 TB/s
//                       for V = 32..128 (k_grad_small: 3.6-6.4), 8.0-8.3 for V = 132..300 (k_grad_span: 5.6-7.3)
//   k_grad_small        dense tensors of up to 128 logits whose rows are no multiple of 16 bytes (V = 62: 6.3;
//                       k_grad_flat<false> 4.6, k_grad 3.0)
//   k_grad_flat<false>  such rows where k_grad_small cannot run (not dense, or more than 128 logits): 4.3-5.0
//   k_grad_span         rows that are multiples of 512 bytes (V = 256: 8.5, 512: 9.1, 1024: 9.3; k_grad_flat 8.1-8.3)
//   k_grad              everything else (unaligned utterances, more than 4096 logits, fewer than 32)
void launch_flat(const Call& c, int b0, int nb, hipStream_t st) {
  FlatGeom fg;
  fg.maxT = c.maxT; fg.maxU1 = c.maxU1; fg.V = c.V; fg.blank = c.blank;
  fg.rows_utt = c.maxT * c.maxU1;
  fg.q256 = 256 / c.V; fg.r256 = 256 % c.V;
  const long long total = (long long)fg.rows_utt * c.V, per_block = 4LL * kFlatChunksPerBlock;
  fg.blocks_utt = (int)((total + per_block - 1) / per_block);
  fg.table_rows = (int)(per_block / c.V) + 2;
  fg.b0 = b0; fg.nblocks = nb * fg.blocks_utt; fg.nblocks_all = c.B * fg.blocks_utt;
  fg.prune_log2 = c.prune_log2;
  fg.xstride = c.xstride; fg.gstride = c.gstride;
  const dim3 grid((unsigned)fg.nblocks), blk(kFlatWaves * 64);
  const size_t lds = sizeof(float4) * (size_t)fg.table_rows;
  if (c.V % 4 == 0)
    k_grad_flat<true><<<grid, blk, lds, st>>>(c.logits, c.grad, c.labels, c.in_len, c.lab_len, c.grad_scale, fg, c.w);
  else
    k_grad_flat<false><<<grid, blk, lds, st>>>(c.logits, c.grad, c.labels, c.in_len, c.lab_len, c.grad_scale, fg, c.w);
}

void zero_cropped_frames(const Call& c, int b0, hipStream_t st) {
  if (c.t_off > 0 && b0 == 0) {
    const size_t n = (size_t)c.t_off * c.maxU1 * c.V;
    const unsigned per_b = (unsigned)((n + 1023) / 1024);
    k_zero_frames<<<dim3(per_b, (unsigned)c.B), 256, 0, st>>>(c.grad - n, c.gstride, n);
  }
}

void launch_grad(const Call& c, int b0, int nb, hipStream_t st) {
  // k_grad_flat only needs every utterance (of the logits and of the gradient) to start on a 16-byte boundary
  const uintptr_t bits = reinterpret_cast<uintptr_t>(c.logits) | reinterpret_cast<uintptr_t>(c.grad) |
                         (uintptr_t)(c.xstride * 4) | (uintptr_t)(c.gstride * 4);
  bool flat_ok = bits % 16 == 0 && c.V >= 32 && c.V <= 4096;
#ifdef RNNT_NO_FLAT
  flat_ok = false;
#endif
  if (flat_ok && c.V % 4 == 0 && (c.V % 128 != 0 || c.V <= 128)) {
    zero_cropped_frames(c, b0, st);
    launch_flat(c, b0, nb, st);
    return;
  }
  if (b0 == 0 && nb == c.B && use_small(c, c.grad)) {
    int blocks;
    const SmallGeom g = small_geom(c, &blocks);
    const size_t lds = sizeof(float4) * (size_t)small_lds_f4(c.V, true) * kSmallWaves;
    const size_t skip = (size_t)c.t_off * c.maxU1 * c.V;  // the cropped frames are rows without a cell: zero-filled
    const float* x0 = c.logits - skip;
    float* g0 = c.grad - skip;
#ifndef RNNT_NO_SMALL_SPAN
    {  // one block per tile (plus one for a partial last tile), when the per-block slots of the workspace suffice
      const long long sblocks = (long long)g.ntiles_full + ((long long)g.ntiles_full * 64 < g.total_rows ? 1 : 0);
      const size_t slds = sizeof(float4) * (size_t)small_lds_f4(c.V, true);
      if (sblocks >= 1 && (size_t)sblocks <= ws_partial_slots(c.B, c.maxT, c.maxU1)) {
        RNNT_SMALL_SPAN_DISPATCH(k_grad_small_span, c.V, true, (int)sblocks, slds, st, x0, g0, c.labels, c.in_len, c.lab_len,
                                 c.grad_scale, g, c.w);
        return;
      }
    }
#endif
    RNNT_SMALL_DISPATCH(k_grad_small, c.V, true, blocks, lds, st, x0, g0, c.labels, c.in_len, c.lab_len, c.grad_scale, g, c.w);
    return;
  }
  zero_cropped_frames(c, b0, st);
  if (flat_ok && c.V % 4 != 0) {
    launch_flat(c, b0, nb, st);
    return;
  }
  const RowCfg cfg = sub_cfg(c, c.grad, b0, nb);
#ifndef RNNT_NO_SPAN
  if (cfg.vec == 4 && c.V > 128 && c.V <= 4096) {
    // rows per wave chosen so that a wave owns about 32 KiB: lanes per row L, 16-byte chunks per lane NC, row groups NG
    SpanGeom sg = span_geom(c, b0);
    const int nchunk = c.V / 4;
    auto go = [&](auto kernel, int rows_per_block) {
      sg.nchunks = (sg.rows_utt + rows_per_block - 1) / rows_per_block;
      sg.nblocks = nb * sg.nchunks;
      sg.nblocks_all = c.B * sg.nchunks;
      kernel<<<dim3((unsigned)sg.nblocks), dim3(kSpanWaves * 64), 0, st>>>(c.logits, c.grad, c.labels, c.in_len, c.lab_len,
                                                                          c.grad_scale, sg, c.w);
    };
#ifndef RNNT_NO_SPLIT
    // rows of whole KiB: classify, zero-fill and read-modify-write as three kernels (grad_split.h) -- unless the
    // caller has switched pruning off: then only padding is zero, and the fill kernel's wavefronts that find nothing
    // to do (one per KiB of the tensor) would cost more than the split gains (measured: 5.75 against 5.2 ms).
    if (c.V % 256 == 0 && c.B <= 65535 && b0 == 0 && nb == c.B && c.prune_log2 > -120.f) {
      SplitGeom sp;
      sp.maxT = c.maxT; sp.maxU1 = c.maxU1; sp.V = c.V; sp.blank = c.blank;
      sp.rows_utt = sg.rows_utt;
      sp.regions_utt = ws_regions(c.maxT, c.maxU1);
      sp.prune_log2 = c.prune_log2;
      sp.xstride = c.xstride; sp.gstride = c.gstride;
      const int ng = nchunk <= 256 ? RNNT_RMW_NG : nchunk <= 512 ? RNNT_RMW_NG / 2 : RNNT_RMW_NG / 4;
      sp.waves = nchunk <= 512 ? kRmwWaves : kRmwWavesWide;
      {  // alpha, beta and the records of the batch against the 256 MiB of the memory-side cache
        const size_t state = (size_t)c.B * ((size_t)c.w.D * c.w.U1s * 16 + (size_t)(c.maxT + 2) * c.maxU1 * 16);
        sp.nt = state > (192u << 20);
#ifdef RNNT_FORCE_NT
        sp.nt = RNNT_FORCE_NT;
#endif
      }
      sp.rb = sp.waves * ng;
      sp.spb = (kRegionRows + sp.rb - 1) / sp.rb;
      const dim3 per_region((unsigned)sp.regions_utt, (unsigned)c.B);
      k_classify_rows<<<per_region, kClassifyBlock, 0, st>>>(c.labels, c.in_len, c.lab_len, c.grad_scale, sp, c.w);
      // a resident grid: two blocks of six wavefronts per compute unit is what their registers allow
      const long long max_items = (long long)c.B * sp.regions_utt * sp.spb;
#ifndef RNNT_RMW_PER_CU  // blocks of the resident grid per compute unit (their registers allow 12 wavefronts)
#define RNNT_RMW_PER_CU (12 / sp.waves)
#endif
      const long long resident = (long long)(RNNT_RMW_PER_CU) * compute_units();
      const unsigned grid_rmw = (unsigned)(max_items < resident ? max_items : resident);
      const dim3 blk(sp.waves * 64);
      auto rmw = [&]() {
        if (nchunk <= 256) k_grad_rmw<4, RNNT_RMW_NG, kRmwWaves><<<grid_rmw, blk, 0, st>>>(c.logits, c.grad, c.grad_scale, sp, c.w);
        else if (nchunk <= 512) k_grad_rmw<8, RNNT_RMW_NG / 2, kRmwWaves><<<grid_rmw, blk, 0, st>>>(c.logits, c.grad, c.grad_scale, sp, c.w);
        else k_grad_rmw<16, RNNT_RMW_NG / 4, kRmwWavesWide><<<grid_rmw, blk, 0, st>>>(c.logits, c.grad, c.grad_scale, sp, c.w);
      };
      const int row_kib = c.V / 256;
      const long long kib_utt = (long long)sg.rows_utt * row_kib;  // < 2^31: rows < 2^31 / B and V <= 4096
      auto fill = [&]() {
#ifndef RNNT_FILL_SPLIT
#define RNNT_FILL_SPLIT 1
#endif
        const int per = (c.B + RNNT_FILL_SPLIT - 1) / RNNT_FILL_SPLIT;
        for (int f0 = 0; f0 < c.B; f0 += per) {
          const int nbf = c.B - f0 < per ? c.B - f0 : per;
          k_fill_rows<<<dim3((unsigned)((kib_utt + kFillWaves - 1) / kFillWaves), (unsigned)nbf), kFillWaves * 64, 0, st>>>(
              c.grad, c.gstride, (int)kib_utt, row_kib, c.w.rowbits, c.w.row_words, f0);
        }
        k_patch_sparse<<<per_region, 256, 0, st>>>(c.grad, c.grad_scale, sp, c.w);
      };
#if defined(RNNT_SPLIT_OVERLAP)  // experiment: the zero-fill on a second stream underneath the read-modify-write kernel
      static hipStream_t side = nullptr;
      static hipEvent_t ev_fork = nullptr, ev_join = nullptr;
      if (!side) {
        (void)hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
        (void)hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&ev_join, hipEventDisableTiming);
      }
      (void)hipEventRecord(ev_fork, st);
      (void)hipStreamWaitEvent(side, ev_fork, 0);
      {
        hipStream_t main_st = st;
        st = side;
        fill();
        st = main_st;
      }
      (void)hipEventRecord(ev_join, side);
      rmw();
      (void)hipStreamWaitEvent(st, ev_join, 0);
#elif defined(RNNT_SPLIT_FILL_FIRST)
      fill();
      rmw();
#else
      rmw();
      fill();
#endif
      return;
    }
#endif
    if (nchunk <= 64) go(k_grad_span<16, 4, RNNT_SPAN_NG>, SpanShape<16, RNNT_SPAN_NG>::ROWS);
    else if (nchunk <= 128) go(k_grad_span<32, 4, RNNT_SPAN_NG>, SpanShape<32, RNNT_SPAN_NG>::ROWS);
    else if (nchunk <= 256) go(k_grad_span<64, 4, RNNT_SPAN_NG>, SpanShape<64, RNNT_SPAN_NG>::ROWS);
    else if (nchunk <= 512) go(k_grad_span<64, 8, RNNT_SPAN_NG / 2>, SpanShape<64, RNNT_SPAN_NG / 2>::ROWS);
    else go(k_grad_span<64, 16, RNNT_SPAN_NG / 4>, SpanShape<64, RNNT_SPAN_NG / 4>::ROWS);
    return;
  }
#endif
  RNNT_DISPATCH(k_grad, cfg, sizeof(float4) * (size_t)cfg.gm.TT * c.maxU1, c.logits, c.grad, c.labels, c.in_len,
                c.lab_len, c.grad_scale, cfg.gm, c.w);
}

int launch_forward(const Call& c, hipStream_t st) {
  k_init<<<1, 64, 0, st>>>(c.w, 0);
  launch_lse(c, 0, c.B, st);
  launch_sweep(c, 0, c.B, st);
  return launch_status();
}

int launch_backward(const Call& c, hipStream_t st) {
  launch_grad(c, 0, c.B, st);
  return launch_status();
}

// forward + backward on the caller's stream.  ev (optional): four hipEvent_t recorded before k_lse, between k_lse
// and k_sweep, between k_sweep and k_grad, and after k_grad; null entries are skipped.  (Running the
// latency-bound sweep of one half of the batch on a second stream underneath the streaming kernels of the other
// half was measured and rejected: the sweep is hidden, but two extra kernel tails and its interference cost as
// much as the 0.1 ms it saves.)
int launch_loss_grad(const Call& c, hipStream_t st, void* const* ev) {
  auto mark = [&](int i) {
    return !ev || !ev[i] || hipEventRecord(static_cast<hipEvent_t>(ev[i]), st) == hipSuccess;
  };
  k_init<<<1, 64, 0, st>>>(c.w, 0);
  if (!mark(0)) return RNNT_MI355X_LAUNCH_FAILURE;
#ifndef RNNT_LSE_SPLIT
#define RNNT_LSE_SPLIT 1
#endif
  if (RNNT_LSE_SPLIT > 1 && !use_small(c, c.logits)) {
    const int per = (c.B + RNNT_LSE_SPLIT - 1) / RNNT_LSE_SPLIT;
    for (int l0 = 0; l0 < c.B; l0 += per) launch_lse(c, l0, c.B - l0 < per ? c.B - l0 : per, st);
  } else
  launch_lse(c, 0, c.B, st);
  if (!mark(1)) return RNNT_MI355X_LAUNCH_FAILURE;
  launch_sweep(c, 0, c.B, st);
  if (!mark(2)) return RNNT_MI355X_LAUNCH_FAILURE;
  int rc = launch_status();
  if (rc != RNNT_MI355X_OK) return rc;
  rc = launch_backward(c, st);
  if (!mark(3)) return RNNT_MI355X_LAUNCH_FAILURE;
  return rc;
}

int make_call(Call* c, const float* logits, float* grad, const int32_t* labels, const int32_t* in_len,
              const int32_t* lab_len, int B, int maxT, int maxU1, int V, int blank, const float* grad_scale,
              float* costs, void* workspace, bool need_grad, bool need_costs, float prune_log2, size_t xstride = 0,
              size_t gstride = 0, int t_off = 0) {
  int rc = check_shape(B, maxT, maxU1, V, blank);
  if (rc != RNNT_MI355X_OK) return rc;
  if (need_grad) {
    rc = resolve_prune(prune_log2, &c->prune_log2);
    if (rc != RNNT_MI355X_OK) return rc;
  }
  c->t_off = t_off;
  const size_t dense = (size_t)maxT * maxU1 * V;
  if ((xstride && xstride < dense) || (gstride && gstride < dense)) return RNNT_MI355X_INVALID_VALUE;
  c->xstride = xstride ? xstride : dense;
  c->gstride = gstride ? gstride : dense;
  if (!logits || !in_len || !lab_len || !workspace || (maxU1 > 1 && !labels)) return RNNT_MI355X_INVALID_VALUE;
  if ((need_grad && !grad) || (need_costs && !costs)) return RNNT_MI355X_INVALID_VALUE;
  c->logits = logits; c->grad = grad; c->labels = labels; c->in_len = in_len; c->lab_len = lab_len;
  c->B = B; c->maxT = maxT; c->maxU1 = maxU1; c->V = V; c->blank = blank;
  c->grad_scale = grad_scale; c->costs = costs;
  ws_layout(workspace, B, maxT, maxU1, &c->w);
  return RNNT_MI355X_OK;
}

// ---- the entry point that allocates nothing ----
// Its workspace is this array of the code object instead of a caller's buffer: every GPU the library runs on holds
// one copy for the life of the process.  `scratch` itself is a host-side stand-in; the address the kernels get is
// the one hipGetSymbolAddress reports for the current device, looked up once per device.
__device__ __align__(256) unsigned char scratch[RNNT_MI355X_STATIC_WORKSPACE_BYTES];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxDevices] = {};

int scratch_of_current_device(void** out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return RNNT_MI355X_LAUNCH_FAILURE;
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return RNNT_MI355X_LAUNCH_FAILURE;
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return RNNT_MI355X_OK;
}

// ---- fused additive joint: host side ----
struct JointCall {
  const float* f;
  const float* g;
  const int32_t* labels;
  const int32_t* in_len;
  const int32_t* lab_len;
  JointGeom jg;
  Ws w;
};

int joint_ke(int V) {
  int ke = 1;
  while (ke * 64 < V) ke <<= 1;
  return ke;  // 1..64 for V <= 4096
}

int make_joint_call(JointCall* c, const float* f, size_t fstride, const float* g, const int32_t* labels,
                    const int32_t* in_len, const int32_t* lab_len, int B, int maxT, int maxU1, int V, int blank,
                    size_t dfstride, void* workspace) {
  const int rc = check_shape(B, maxT, maxU1, V, blank);
  if (rc != RNNT_MI355X_OK) return rc;
  if (V > 4096) return RNNT_MI355X_UNSUPPORTED_SIZE;
  if (!f || !g || !in_len || !lab_len || !workspace || (maxU1 > 1 && !labels)) return RNNT_MI355X_INVALID_VALUE;
  const size_t dense = (size_t)maxT * V;
  if ((fstride && fstride < dense) || (dfstride && dfstride < dense)) return RNNT_MI355X_INVALID_VALUE;
  c->f = f; c->g = g; c->labels = labels; c->in_len = in_len; c->lab_len = lab_len;
  c->jg.B = B; c->jg.maxT = maxT; c->jg.maxU1 = maxU1; c->jg.V = V; c->jg.blank = blank;
  c->jg.fstride = fstride ? fstride : dense;
  c->jg.dfstride = dfstride ? dfstride : dense;
  ws_layout(workspace, B, maxT, maxU1, &c->w, true, V);
  return RNNT_MI355X_OK;
}

#define RNNT_JOINT_DISPATCH(KERNEL, ke, GRID, T16, T32, T64, ...)                       \
  do {                                                                                  \
    switch (ke) {                                                                       \
      case 1: KERNEL<1, 4><<<GRID(4), 256, 0, st>>>(__VA_ARGS__); break;                \
      case 2: KERNEL<2, 4><<<GRID(4), 256, 0, st>>>(__VA_ARGS__); break;                \
      case 4: KERNEL<4, 4><<<GRID(4), 256, 0, st>>>(__VA_ARGS__); break;                \
      case 8: KERNEL<8, 4><<<GRID(4), 256, 0, st>>>(__VA_ARGS__); break;                \
      case 16: KERNEL<16, T16><<<GRID(T16), 256, 0, st>>>(__VA_ARGS__); break;          \
      case 32: KERNEL<32, T32><<<GRID(T32), 256, 0, st>>>(__VA_ARGS__); break;          \
      default: KERNEL<64, T64><<<GRID(T64), 256, 0, st>>>(__VA_ARGS__); break;          \
    }                                                                                   \
  } while (0)

// blocks of four wavefronts for a batched contraction with m_tiles x n_strips wavefront tiles per utterance
inline unsigned mm_blocks(int B, int m_tiles, int n_strips) {
  return (unsigned)(((long long)B * m_tiles * n_strips + 3) / 4);
}

int launch_joint_forward(const JointCall& c, float* costs, hipStream_t st) {
  const JointGeom& jg = c.jg;
  const Ws& w = c.w;
  const int groups = jg.B * ((w.Tp + w.U1p) / 4);  // k_joint_prep also starts the workspace header
  const bool rows16 = jg.V % 4 == 0 && jg.fstride % 4 == 0 && reinterpret_cast<uintptr_t>(c.f) % 16 == 0 &&
                      reinterpret_cast<uintptr_t>(c.g) % 16 == 0;
  const unsigned pgrid = (unsigned)((groups + 3) / 4);
#define JOINT_PREP(NIT)                                                            \
  do {                                                                             \
    if (rows16) k_joint_prep<true, NIT><<<pgrid, 256, 0, st>>>(c.f, c.g, c.labels, jg, w);   \
    else k_joint_prep<false, NIT><<<pgrid, 256, 0, st>>>(c.f, c.g, c.labels, jg, w);         \
  } while (0)
  if (w.Vp <= 256) JOINT_PREP(4);
  else if (w.Vp <= 1024) JOINT_PREP(16);
  else if (w.Vp <= 2048) JOINT_PREP(32);
  else JOINT_PREP(0);
#undef JOINT_PREP
  const long long ncell = (long long)jg.B * jg.maxT * jg.maxU1;
  k_joint_gather<<<(unsigned)((ncell + 255) / 256), 256, 0, st>>>(c.f, c.labels, c.in_len, c.lab_len, jg, w);
  const int mt = w.Tp / 32;
  if (w.U1p <= 32)
    k_joint_lse_mm<1><<<mm_blocks(jg.B, mt, 1), 256, 0, st>>>(c.f, c.g, c.in_len, c.lab_len, jg, w);
  // strips of 128 label positions where the sum over the vocabulary is long: fewer operand loads per MFMA (T=400,
  // U+1=101, V=1024: 89 -> 83 us; at V=62, where a wave's whole sum is 4 steps, the halved wave count loses: 12.7 -> 16.6 us)
  else if (w.U1p > 64 && w.Vp >= 256)
    k_joint_lse_mm<4><<<mm_blocks(jg.B, mt, (w.U1p + 127) / 128), 256, 0, st>>>(c.f, c.g, c.in_len, c.lab_len, jg, w);
  else
    k_joint_lse_mm<2><<<mm_blocks(jg.B, mt, (w.U1p + 63) / 64), 256, 0, st>>>(c.f, c.g, c.in_len, c.lab_len, jg, w);
  Call sc;  // the lattice sweep is the one of the materialised path
  sc.in_len = c.in_len; sc.lab_len = c.lab_len; sc.B = jg.B; sc.maxT = jg.maxT; sc.maxU1 = jg.maxU1;
  sc.costs = costs; sc.w = w;
  launch_sweep(sc, 0, jg.B, st);
  return launch_status();
}

int launch_joint_backward(const JointCall& c, const float* grad_scale, float* df, float* dg, hipStream_t st) {
  const JointGeom& jg = c.jg;
  const Ws& w = c.w;
  const int ke = joint_ke(jg.V);
  // Large vocabularies: few repeats per symbol, a big d f -> the blank / label terms go into the epilogue of the
  // d f contraction.  Small ones (the 40 / 62-symbol sets): long chains, a small d f -> a pass of their own.
  const bool fold = w.Vp >= 256;
  if (fold)
    k_joint_labelmap<<<jg.B, 256, sizeof(int) * (size_t)(jg.maxU1 + 2), st>>>(c.labels, c.in_len, c.lab_len, jg, w);
  k_joint_coef<<<(unsigned)(jg.B * (w.Tp / 8) * (w.U1p / 32)), 256, 0, st>>>(c.labels, c.in_len, c.lab_len, grad_scale,
                                                                             jg, w);
  const int mt = w.Tp / 32, mu = w.U1p / 32;
  const int nt = w.Vp <= 32 ? 1 : w.Vp <= 64 ? 2 : 4;
  // results may leave as vectors of nt floats when every row of d f / d g starts on a multiple of nt floats
  auto aligned = [&](const float* p, size_t stride) {
    return jg.V % nt == 0 && stride % nt == 0 && reinterpret_cast<uintptr_t>(p) % (sizeof(float) * nt) == 0;
  };
  const bool vf = aligned(df, jg.dfstride), vg = aligned(dg, (size_t)nt);
  const int ns = w.Vp / (32 * nt);
  if (nt == 1) {
    k_joint_df_mm<1, false><<<mm_blocks(jg.B, mt, ns), 256, 0, st>>>(df, c.in_len, c.lab_len, jg, w, vf);
    k_joint_dg_mm<1><<<mm_blocks(jg.B * w.dg_split, mu, ns), 256, 0, st>>>(dg, c.in_len, c.lab_len, jg, w, vg);
  } else if (nt == 2) {
    k_joint_df_mm<2, false><<<mm_blocks(jg.B, mt, ns), 256, 0, st>>>(df, c.in_len, c.lab_len, jg, w, vf);
    k_joint_dg_mm<2><<<mm_blocks(jg.B * w.dg_split, mu, ns), 256, 0, st>>>(dg, c.in_len, c.lab_len, jg, w, vg);
  } else {
    if (fold) k_joint_df_mm<4, true><<<mm_blocks(jg.B, mt, ns), 256, 0, st>>>(df, c.in_len, c.lab_len, jg, w, vf);
    else k_joint_df_mm<4, false><<<mm_blocks(jg.B, mt, ns), 256, 0, st>>>(df, c.in_len, c.lab_len, jg, w, vf);
#ifndef RNNT_JOINT_NO_LDS
    if (mu % 4 == 0) {  // whole blocks of four row tiles: the vocabulary-side operand goes through LDS once per block
      const unsigned dg_blocks = (unsigned)((long long)jg.B * w.dg_split * ns * (mu / 4));
      k_joint_dg_mm_lds<<<dg_blocks, 256, 0, st>>>(dg, c.in_len, c.lab_len, jg, w, vg);
    } else
#endif
    k_joint_dg_mm<4><<<mm_blocks(jg.B * w.dg_split, mu, ns), 256, 0, st>>>(dg, c.in_len, c.lab_len, jg, w, vg);
  }
  if (!fold) {
    // frames per block: (3 + fb) * (maxU1 + 1) words of LDS.  8 measured best at T=500, U=100, V=62 (18 us; 16: 20 us,
    // 32: 26 us): more blocks outweigh building the chains more often.
    const int fb = 8;
    const size_t corr_lds = sizeof(int) * (3 + fb) * (size_t)(jg.maxU1 + 1);
    k_joint_corr_f<<<jg.B * ((jg.maxT + fb - 1) / fb), 256, corr_lds, st>>>(df, c.labels, c.in_len, c.lab_len, jg, w, fb);
  }
  k_joint_corr_g<<<jg.B * ((jg.maxU1 + 63) / 64), 1024, 0, st>>>(dg, c.labels, c.in_len, c.lab_len, jg, w);
#define JBLOCKS_T(TT) (jg.B * ((((jg.maxT + (TT)-1) / (TT)) + 3) / 4))
#define JBLOCKS_U(UU) (jg.B * ((((jg.maxU1 + (UU)-1) / (UU)) + 3) / 4))
#define JGRID_FIX(TT) dim3((unsigned)(JBLOCKS_T(TT) + JBLOCKS_U(TT)))
  const int fb4 = JBLOCKS_T(4), fb2 = JBLOCKS_T(2), fb1 = JBLOCKS_T(1);
  const int f_blocks = ke <= 8 ? fb4 : ke == 16 ? fb2 : fb1;  // the tile heights RNNT_JOINT_DISPATCH picks below
  RNNT_JOINT_DISPATCH(k_joint_fix, ke, JGRID_FIX, 2, 1, 1, c.f, c.g, df, dg, c.in_len, c.lab_len, grad_scale, jg, w,
                      f_blocks);
  return launch_status();
}

}  // namespace

extern "C" {

// Timing-experiment builds that knock a piece of a kernel out (and so compute wrong results) report a negative
// version: the Python loader refuses them unless the experiment asks for them by name.
#if defined(RNNT_SPAN_FAKE_PROLOGUE) || defined(RNNT_RMW_FAKE_CONTIG) || defined(RNNT_DENSE_NO_LOAD) || \
    defined(RNNT_DENSE_NO_MFMA) || defined(RNNT_DENSE_NO_STORE)
int rnnt_mi355x_version(void) { return -300; }
#else
int rnnt_mi355x_version(void) { return 300; }
#endif

#ifdef RNNT_RMW_TIMING
// experiment builds only: reads and clears the per-phase cycle sums of k_grad_rmw
int rnnt_mi355x_debug_rmw_timing(unsigned long long* host8) {
  if (hipDeviceSynchronize() != hipSuccess) return RNNT_MI355X_LAUNCH_FAILURE;
  if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_rmw_dbg), 64) != hipSuccess) return RNNT_MI355X_LAUNCH_FAILURE;
  unsigned long long z[8] = {};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_rmw_dbg), z, 64) != hipSuccess) return RNNT_MI355X_LAUNCH_FAILURE;
  return RNNT_MI355X_OK;
}
// experiment builds only: k_fill_rows alone (rowbits: B * row_words words, row_words = ceil(rows / 64) * 2)
int rnnt_mi355x_debug_fill(float* grad, size_t gstride, int rows_utt, int row_kib, const uint32_t* rowbits, int B,
                           void* stream) {
  const long long kib_utt = (long long)rows_utt * row_kib;
  const int row_words = (rows_utt + 63) / 64 * 2;
  k_fill_rows<<<dim3((unsigned)((kib_utt + kFillWaves - 1) / kFillWaves), (unsigned)B), kFillWaves * 64, 0,
                static_cast<hipStream_t>(stream)>>>(grad, gstride, (int)kib_utt, row_kib, rowbits, row_words, 0);
  return launch_status();
}
int rnnt_mi355x_debug_rmw_spans(unsigned long long* host2048) {
  if (hipDeviceSynchronize() != hipSuccess) return RNNT_MI355X_LAUNCH_FAILURE;
  if (hipMemcpyFromSymbol(host2048, HIP_SYMBOL(g_rmw_span), 2 * 1024 * 8) != hipSuccess) return RNNT_MI355X_LAUNCH_FAILURE;
  return RNNT_MI355X_OK;
}
#endif

int rnnt_mi355x_set_prune_log2(float log2_threshold) {
  if (!(log2_threshold <= -16.f)) return RNNT_MI355X_INVALID_VALUE;  // also rejects NaN
  g_prune_log2.store(log2_threshold, std::memory_order_relaxed);
  return RNNT_MI355X_OK;
}

float rnnt_mi355x_get_prune_log2(void) { return g_prune_log2.load(std::memory_order_relaxed); }

const char* rnnt_mi355x_status_string(int status) {
  switch (status) {
    case RNNT_MI355X_OK: return "ok";
    case RNNT_MI355X_INVALID_VALUE: return "invalid value (null pointer, non-positive size or blank outside [0,V))";
    case RNNT_MI355X_UNSUPPORTED_SIZE: return "unsupported size (beyond the limits stated in rnnt_mi355x.h)";
    case RNNT_MI355X_LAUNCH_FAILURE: return "HIP launch or copy failure";
    case RNNT_MI355X_BAD_INPUT_DATA: return "device-side validation failed (length or label out of range)";
    default: return "unknown status";
  }
}

int rnnt_mi355x_workspace_bytes(int B, int maxT, int maxU1, size_t* out_bytes) {
  if (!out_bytes) return RNNT_MI355X_INVALID_VALUE;
  const int rc = check_shape(B, maxT, maxU1, 1, 0);
  if (rc != RNNT_MI355X_OK) return rc;
  *out_bytes = ws_layout(nullptr, B, maxT, maxU1, nullptr);
  return RNNT_MI355X_OK;
}

int rnnt_mi355x_forward(const float* logits, const int32_t* labels, const int32_t* in_len, const int32_t* lab_len,
                        int B, int maxT, int maxU1, int V, int blank, float* costs, void* workspace,
                        void* hip_stream) {
  Call c;
  const int rc = make_call(&c, logits, nullptr, labels, in_len, lab_len, B, maxT, maxU1, V, blank, nullptr, costs,
                           workspace, false, true, 0.f);
  if (rc != RNNT_MI355X_OK) return rc;
  return launch_forward(c, static_cast<hipStream_t>(hip_stream));
}

int rnnt_mi355x_backward(const float* logits, float* grad, const int32_t* labels, const int32_t* in_len,
                         const int32_t* lab_len, int B, int maxT, int maxU1, int V, int blank, const float* grad_scale,
                         void* workspace, void* hip_stream, float prune_log2) {
  Call c;
  const int rc = make_call(&c, logits, grad, labels, in_len, lab_len, B, maxT, maxU1, V, blank, grad_scale, nullptr,
                           workspace, true, false, prune_log2);
  if (rc != RNNT_MI355X_OK) return rc;
  return launch_backward(c, static_cast<hipStream_t>(hip_stream));
}

int rnnt_mi355x_forward_strided(const float* logits, size_t logits_batch_stride, const int32_t* labels,
                                const int32_t* in_len, const int32_t* lab_len, int B, int maxT, int maxU1, int V,
                                int blank, float* costs, void* workspace, void* hip_stream) {
  Call c;
  const int rc = make_call(&c, logits, nullptr, labels, in_len, lab_len, B, maxT, maxU1, V, blank, nullptr, costs,
                           workspace, false, true, 0.f, logits_batch_stride, 0);
  if (rc != RNNT_MI355X_OK) return rc;
  return launch_forward(c, static_cast<hipStream_t>(hip_stream));
}

int rnnt_mi355x_backward_strided(const float* logits, size_t logits_batch_stride, float* grad,
                                 size_t grad_batch_stride, const int32_t* labels, const int32_t* in_len,
                                 const int32_t* lab_len, int B, int maxT, int maxU1, int V, int blank,
                                 const float* grad_scale, void* workspace, void* hip_stream, float prune_log2) {
  Call c;
  const int rc = make_call(&c, logits, grad, labels, in_len, lab_len, B, maxT, maxU1, V, blank, grad_scale, nullptr,
                           workspace, true, false, prune_log2, logits_batch_stride, grad_batch_stride);
  if (rc != RNNT_MI355X_OK) return rc;
  return launch_backward(c, static_cast<hipStream_t>(hip_stream));
}

int rnnt_mi355x_forward_cropped(const float* logits_full, const int32_t* labels, const int32_t* in_len,
                                const int32_t* lab_len, int B, int T0, int t_offset, int maxU1, int V, int blank,
                                float* costs, void* workspace, void* hip_stream) {
  if (t_offset < 0 || t_offset >= T0 || !logits_full) return RNNT_MI355X_INVALID_VALUE;
  if ((long long)B * T0 * maxU1 >= (1LL << 31)) return RNNT_MI355X_UNSUPPORTED_SIZE;
  const size_t frame = (size_t)maxU1 * V, stride = (size_t)T0 * frame;
  Call c;
  const int rc = make_call(&c, logits_full + t_offset * frame, nullptr, labels, in_len, lab_len, B, T0 - t_offset,
                           maxU1, V, blank, nullptr, costs, workspace, false, true, 0.f, stride, stride, t_offset);
  if (rc != RNNT_MI355X_OK) return rc;
  return launch_forward(c, static_cast<hipStream_t>(hip_stream));
}

int rnnt_mi355x_backward_cropped(const float* logits_full, float* grad_full, const int32_t* labels,
                                 const int32_t* in_len, const int32_t* lab_len, int B, int T0, int t_offset,
                                 int maxU1, int V, int blank, const float* grad_scale, void* workspace,
                                 void* hip_stream, float prune_log2) {
  if (t_offset < 0 || t_offset >= T0 || !logits_full || !grad_full) return RNNT_MI355X_INVALID_VALUE;
  if ((long long)B * T0 * maxU1 >= (1LL << 31)) return RNNT_MI355X_UNSUPPORTED_SIZE;
  const size_t frame = (size_t)maxU1 * V, stride = (size_t)T0 * frame;
  Call c;
  const int rc = make_call(&c, logits_full + t_offset * frame, grad_full + t_offset * frame, labels, in_len, lab_len,
                           B, T0 - t_offset, maxU1, V, blank, grad_scale, nullptr, workspace, true, false, prune_log2,
                           stride, stride, t_offset);
  if (rc != RNNT_MI355X_OK) return rc;
  return launch_backward(c, static_cast<hipStream_t>(hip_stream));
}

int rnnt_mi355x_backward_inplace(float* logits_inout, const int32_t* labels, const int32_t* in_len,
                                 const int32_t* lab_len, int B, int maxT, int maxU1, int V, int blank,
                                 const float* grad_scale, void* workspace, void* hip_stream, float prune_log2) {
  return rnnt_mi355x_backward(logits_inout, logits_inout, labels, in_len, lab_len, B, maxT, maxU1, V, blank,
                              grad_scale, workspace, hip_stream, prune_log2);
}

int rnnt_mi355x_loss_grad(float* logits_inout, const int32_t* labels, const int32_t* in_len, const int32_t* lab_len,
                          int B, int maxT, int maxU1, int V, int blank, const float* grad_scale, float* costs,
                          void* workspace, void* hip_stream, float prune_log2) {
  return rnnt_mi355x_loss_grad_events(logits_inout, labels, in_len, lab_len, B, maxT, maxU1, V, blank, grad_scale,
                                      costs, workspace, hip_stream, prune_log2, nullptr);
}

int rnnt_mi355x_loss_grad_events(float* logits_inout, const int32_t* labels, const int32_t* in_len,
                                 const int32_t* lab_len, int B, int maxT, int maxU1, int V, int blank,
                                 const float* grad_scale, float* costs, void* workspace, void* hip_stream,
                                 float prune_log2, void* const* events4) {
  Call c;
  const int rc = make_call(&c, logits_inout, logits_inout, labels, in_len, lab_len, B, maxT, maxU1, V, blank,
                           grad_scale, costs, workspace, true, true, prune_log2);
  if (rc != RNNT_MI355X_OK) return rc;
  return launch_loss_grad(c, static_cast<hipStream_t>(hip_stream), events4);
}

int rnnt_mi355x_loss_grad_static(float* logits_inout, const int32_t* labels, const int32_t* in_len,
                                 const int32_t* lab_len, int B, int maxT, int maxU1, int V, int blank,
                                 const float* grad_scale, float* costs, double* scalars2, int32_t* flags_out,
                                 void* hip_stream, float prune_log2) {
  int rc = check_shape(B, maxT, maxU1, V, blank);
  if (rc != RNNT_MI355X_OK) return rc;
  if (ws_layout(nullptr, B, maxT, maxU1, nullptr) > (size_t)RNNT_MI355X_STATIC_WORKSPACE_BYTES)
    return RNNT_MI355X_UNSUPPORTED_SIZE;
  void* ws = nullptr;
  rc = scratch_of_current_device(&ws);
  if (rc != RNNT_MI355X_OK) return rc;
  Call c;
  rc = make_call(&c, logits_inout, logits_inout, labels, in_len, lab_len, B, maxT, maxU1, V, blank, grad_scale, costs,
                 ws, true, true, prune_log2);
  if (rc != RNNT_MI355X_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  rc = launch_loss_grad(c, st, nullptr);
  if (rc != RNNT_MI355X_OK) return rc;
  if (scalars2) {
    k_reduce_scalars<<<1, 1024, 0, st>>>(costs, B, static_cast<const char*>(ws), scalars2);
    rc = launch_status();
    if (rc != RNNT_MI355X_OK) return rc;
  }
  if (flags_out && hipMemcpyAsync(flags_out, ws, sizeof(int32_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
    return RNNT_MI355X_LAUNCH_FAILURE;
  return RNNT_MI355X_OK;
}

int rnnt_mi355x_joint_workspace_bytes(int B, int maxT, int maxU1, int V, size_t* out_bytes) {
  if (!out_bytes) return RNNT_MI355X_INVALID_VALUE;
  const int rc = check_shape(B, maxT, maxU1, V, 0);
  if (rc != RNNT_MI355X_OK) return rc;
  if (V > 4096) return RNNT_MI355X_UNSUPPORTED_SIZE;
  *out_bytes = ws_layout(nullptr, B, maxT, maxU1, nullptr, true, V);
  return RNNT_MI355X_OK;
}

int rnnt_mi355x_joint_forward(const float* f, size_t f_batch_stride, const float* g, const int32_t* labels,
                              const int32_t* in_len, const int32_t* lab_len, int B, int maxT, int maxU1, int V,
                              int blank, float* costs, void* workspace, void* hip_stream) {
  JointCall c;
  const int rc = make_joint_call(&c, f, f_batch_stride, g, labels, in_len, lab_len, B, maxT, maxU1, V, blank, 0,
                                 workspace);
  if (rc != RNNT_MI355X_OK) return rc;
  if (!costs) return RNNT_MI355X_INVALID_VALUE;
  return launch_joint_forward(c, costs, static_cast<hipStream_t>(hip_stream));
}

int rnnt_mi355x_joint_backward(const float* f, size_t f_batch_stride, const float* g, const int32_t* labels,
                               const int32_t* in_len, const int32_t* lab_len, int B, int maxT, int maxU1, int V,
                               int blank, const float* grad_scale, float* df, size_t df_batch_stride, float* dg,
                               void* workspace, void* hip_stream, float prune_log2) {
  JointCall c;
  int rc = make_joint_call(&c, f, f_batch_stride, g, labels, in_len, lab_len, B, maxT, maxU1, V, blank,
                           df_batch_stride, workspace);
  if (rc != RNNT_MI355X_OK) return rc;
  if (!df || !dg) return RNNT_MI355X_INVALID_VALUE;
  rc = resolve_prune(prune_log2, &c.jg.prune_log2);
  if (rc != RNNT_MI355X_OK) return rc;
  return launch_joint_backward(c, grad_scale, df, dg, static_cast<hipStream_t>(hip_stream));
}

int rnnt_mi355x_reduce_scalars(const float* costs, int B, void* workspace, double* out2, void* hip_stream) {
  if (!costs || !workspace || !out2 || B < 1) return RNNT_MI355X_INVALID_VALUE;
  k_reduce_scalars<<<1, 1024, 0, static_cast<hipStream_t>(hip_stream)>>>(costs, B, static_cast<const char*>(workspace),
                                                                        out2);
  return launch_status();
}

int rnnt_mi355x_read_stats(void* workspace, uint64_t* host_cells, uint64_t* host_pruned, void* hip_stream) {
  if (!workspace || !host_cells || !host_pruned) return RNNT_MI355X_INVALID_VALUE;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  k_reduce_counts<<<1, 1024, 0, st>>>(static_cast<char*>(workspace));
  if (launch_status() != RNNT_MI355X_OK) return RNNT_MI355X_LAUNCH_FAILURE;
  uint64_t two[2] = {0, 0};
  if (hipMemcpyAsync(two, static_cast<const char*>(workspace) + offsetof(Header, n_cells), sizeof(two),
                     hipMemcpyDeviceToHost, st) != hipSuccess)
    return RNNT_MI355X_LAUNCH_FAILURE;
  if (hipStreamSynchronize(st) != hipSuccess) return RNNT_MI355X_LAUNCH_FAILURE;
  *host_cells = two[0];
  *host_pruned = two[1];
  return RNNT_MI355X_OK;
}

int rnnt_mi355x_read_status(const void* workspace, int32_t* host_flags, void* hip_stream) {
  if (!workspace || !host_flags) return RNNT_MI355X_INVALID_VALUE;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  int32_t flags = 0;
  if (hipMemcpyAsync(&flags, workspace, sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess)
    return RNNT_MI355X_LAUNCH_FAILURE;
  if (hipStreamSynchronize(st) != hipSuccess) return RNNT_MI355X_LAUNCH_FAILURE;
  *host_flags = flags;
  return flags ? RNNT_MI355X_BAD_INPUT_DATA : RNNT_MI355X_OK;
}

}  // extern "C"
