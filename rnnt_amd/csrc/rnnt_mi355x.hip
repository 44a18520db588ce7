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
// Lattice cells whose occupancy gamma = exp(alpha+beta-ll) is below 2^kPruneLog2 (~9e-13) get an exactly zero
// gradient row: every element of that row is bounded by gamma, far below float32 resolution of any sum it
// could enter.  Their logits are then not read a second time.
constexpr float kPruneLog2 = -40.f;

// First bytes of the workspace, written on the device.
struct Header {
  int32_t flags;          // OR of RNNT_MI355X_FLAG_*
  int32_t n_partial;      // entries of `partial` the last k_grad wrote
  uint64_t partial_off;   // byte offset of `partial` from the workspace base
  int32_t n_fallback;     // fused joint: lattice cells whose linear-domain sum underflowed (handled by the exp path)
  int32_t pad[59];
};
static_assert(sizeof(Header) == 256, "header is one 256-byte block");

struct Ws {
  Header* hdr;
  double* ll;      // [2][B]  log2-likelihood from alpha / from beta
  float* partial;  // [B*maxT] per-block sum of squared gradient elements
  // [B][maxT+2][maxU1] {log2 P(blank|t,u), log2 P(label u+1|t,u), log2-sum-exp2 of the row, 0}; frame t lives at
  // index t+1, frames -1 and maxT are guards.  Every entry that is not a lattice cell (guards, t >= T_b,
  // u > U_b) and the label slot of column U_b hold "log 0", so the sweeps need no masks.
  float4* rec;
  double* alpha;   // [B][D][U1s]
  double* beta;    // [B][D][U1s]
  // fused additive joint only:
  float* mf2;      // [B][maxT]  max_v f(b,t,v) * log2(e)
  float* mg2;      // [B][maxU1] max_v g(b,u,v) * log2(e)
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
  int Vp, U1p, Tp; // padded extents: maxT and maxU1 rounded up to 32, V to 32 / 64 / a multiple of 128
  size_t partial_off;
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
  const size_t o_partial = take(sizeof(float) * (size_t)B * maxT);
  const size_t o_rec = take(sizeof(float4) * (size_t)B * (maxT + 2) * maxU1);
  const size_t o_alpha = take(sizeof(double) * plane);
  const size_t o_beta = take(sizeof(double) * plane);
  const size_t o_mf2 = joint ? take(sizeof(float) * (size_t)B * maxT) : 0;
  const size_t o_mg2 = joint ? take(sizeof(float) * (size_t)B * maxU1) : 0;
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
  if (w) {
    w->hdr = reinterpret_cast<Header*>(p + o_hdr);
    w->ll = reinterpret_cast<double*>(p + o_ll);
    w->partial = reinterpret_cast<float*>(p + o_partial);
    w->rec = reinterpret_cast<float4*>(p + o_rec);
    w->alpha = reinterpret_cast<double*>(p + o_alpha);
    w->beta = reinterpret_cast<double*>(p + o_beta);
    w->mf2 = joint ? reinterpret_cast<float*>(p + o_mf2) : nullptr;
    w->mg2 = joint ? reinterpret_cast<float*>(p + o_mg2) : nullptr;
    w->coef = joint ? reinterpret_cast<float4*>(p + o_coef) : nullptr;
    w->Fp = joint ? reinterpret_cast<float*>(p + o_Fp) : nullptr;
    w->Gp = joint ? reinterpret_cast<float*>(p + o_Gp) : nullptr;
    w->Ctu = joint ? reinterpret_cast<float*>(p + o_Ctu) : nullptr;
    w->Cut = joint ? reinterpret_cast<float*>(p + o_Cut) : nullptr;
    w->head_of = joint ? reinterpret_cast<int*>(p + o_head) : nullptr;
    w->nxt = joint ? reinterpret_cast<int*>(p + o_nxt) : nullptr;
    w->rowpart = joint ? reinterpret_cast<float*>(p + o_rowpart) : nullptr;
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
  size_t xstride;   // elements between consecutive utterances of the logits (>= maxT*maxU1*V)
  size_t gstride;   // same for the gradient buffer
  int nblocks;      // blocks of this launch = (utterances of this launch) * nTB
  int nblocks_all;  // blocks of the whole batch = B * nTB
};

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

// Lengths of utterance b, or T = 0 (no valid cell) when they are out of range.
__device__ __forceinline__ void utt_dims(const int32_t* in_len, const int32_t* lab_len, int b, int maxT, int maxU1,
                                         int& T, int& U) {
  const int t = in_len[b], u = lab_len[b];
  const bool ok = (t >= 1) & (t <= maxT) & (u >= 0) & (u <= maxU1 - 1);
  T = ok ? t : 0;
  U = ok ? u : 0;
}

// The logits are streamed exactly once per kernel: non-temporal loads and
// stores (measured on MI355X: 7.0 vs 6.3 TB/s for a pure read stream).
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    const f4v q = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  } else if constexpr (VEC == 2) {
    const f2v q = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(p));
    v[0] = q.x; v[1] = q.y;
  } else {
    v[0] = __builtin_nontemporal_load(p);
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    f4v q = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(q, reinterpret_cast<f4v*>(p));
  } else if constexpr (VEC == 2) {
    f2v q = {v[0], v[1]};
    __builtin_nontemporal_store(q, reinterpret_cast<f2v*>(p));
  } else {
    __builtin_nontemporal_store(v[0], p);
  }
}

// All-reduce inside aligned groups of L lanes.  The steps inside a row of 16
// lanes are DPP operand modifiers (quad swaps, half-row mirror, row mirror:
// one VALU instruction each, no LDS); only the 16- and 32-lane steps go
// through the LDS crossbar.  max and + are commutative, so paired lanes end up
// with bit-identical values.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int L, typename Op>
__device__ __forceinline__ float group_reduce(float v, Op op) {
  static_assert(L == 8 || L == 16 || L == 32 || L == 64, "group size");

  v = op(v, dpp_mov<0xB1>(v));   // quad_perm [1,0,3,2]
  v = op(v, dpp_mov<0x4E>(v));   // quad_perm [2,3,0,1]
  v = op(v, dpp_mov<0x141>(v));  // row_half_mirror
  if constexpr (L >= 16) v = op(v, dpp_mov<0x140>(v));  // row_mirror
  if constexpr (L >= 32) v = op(v, __shfl_xor(v, 16, 64));
  if constexpr (L >= 64) v = op(v, __shfl_xor(v, 32, 64));
  return v;
}
template <int L>
__device__ __forceinline__ float group_max(float v) {
  return group_reduce<L>(v, [](float a, float b) { return fmaxf(a, b); });
}
template <int L>
__device__ __forceinline__ float group_sum(float v) {
  return group_reduce<L>(v, [](float a, float b) { return a + b; });
}

// Rows of the current tile owned by this lane, advanced incrementally (no
// division in the loop).  Row rho of the block is frame t0 + tt, column uu.
template <int G>
struct RowCursor {
  int rho[G], tt[G], uu[G];
  int pass;
  __device__ __forceinline__ void init(int first_row, int rows_apart, int maxU1) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      rho[g] = first_row + g * rows_apart;
      tt[g] = rho[g] / maxU1;
      uu[g] = rho[g] - tt[g] * maxU1;
    }
    pass = 0;
  }
  __device__ __forceinline__ void advance(int npass, int S, int sq, int sr, int maxU1) {
    if (++pass == npass) {
      pass = 0;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        rho[g] += S;
        tt[g] += sq;
        uu[g] += sr;
        if (uu[g] >= maxU1) {
          uu[g] -= maxU1;
          tt[g] += 1;
        }
      }
    }
  }
};

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
        const float lse2 = m_run[g] * kLog2e + fast_log2(s_all);
        if (tl.rho[g] >= 0 && j == 0)
          recs[tl.rho[g]] = make_float4(__builtin_fmaf(exb[g], kLog2e, -lse2),
                                        tl.haslab[g] ? __builtin_fmaf(exl[g], kLog2e, -lse2) : kNegF, lse2, 0.f);
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

// ---------------------------------------------------------------------------
// k_sweep: alpha (even blocks) and beta (odd blocks) anti-diagonal sweeps.
// One wavefront per (utterance, direction); lane l owns the K lattice columns
// u = l*K .. l*K+K-1, so one cross-lane shift (DPP, no LDS) per diagonal is the
// only communication and no barrier sits on the dependent chain.  Cells
// outside the lattice are handled by masking the log-probs that lead into or
// out of them to "log 0": their alpha/beta then evaluate to ~-1e30 by
// themselves.  The log-prob rows of the next P diagonals are kept in flight in
// a register ring.  A diagonal costs the latency of one dependent chain
// (add, shift, subtract, exp2, log2, add), not issue slots: splitting the
// columns of a direction over two wavefronts (one column per lane, boundary
// value through LDS two diagonals ahead, one s_barrier per diagonal) measured
// 2.5x slower, 207 us against 83 us at T=400, U=100.  Carrying alpha and beta
// in the linear domain as (float64 mantissa, int32 exponent) pairs -- multiply,
// ldexp, add, frexp instead of exp2/log2 on the chain -- gave the same results
// and was also slower (one column per lane 77 us against 61 us, two columns
// 135 us against 76 us at T=500): the float64 ldexp/frexp steps cost more than
// the transcendentals they replace.  Measured per diagonal: 109 ns with one
// column per lane, 127 ns with two, 230 ns with four, 494 ns with eight.
// ---------------------------------------------------------------------------
// lane i receives lane i-1's value; lane 0 receives `fill`
__device__ __forceinline__ double lane_shift_up(double v, double fill) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, 0x138, 0xf, 0xf, false);  // wave_shr:1
  hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// lane i receives lane i+1's value; lane 63 receives `fill`
__device__ __forceinline__ double lane_shift_down(double v, double fill) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, 0x130, 0xf, 0xf, false);  // wave_shl:1
  hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Log-probs {blank, label} of the cells (d-u, u) of diagonal d for this lane's K columns.  Records are in [t][u]
// order, so the lanes of a diagonal read with a stride of maxU1-1 records; a cache line serves 8 diagonals.
// Frames are clamped into the guard frames (-1 and maxT), columns beyond maxU1-1 read column maxU1-1 of the
// guard frame; everything outside the lattice reads "log 0".  rec0 = record of frame -1, column 0.
template <int K>
__device__ __forceinline__ void load_lp_diag(const float4* rec0, int d, int maxT, int maxU1, const int (&ucol)[K],
                                             const int (&uoff)[K], float (&pb)[K], float (&pl)[K]) {
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int t1 = min(max(d - ucol[i], -1), maxT) + 1;                    // frame index incl. guards (v_med3_i32)
    // 24-bit multiply-add (full rate; t1 <= maxT + 1 < 2^24, maxU1 <= 1024), 32-bit byte offset from a uniform base
    const unsigned off = __umul24((unsigned)t1, (unsigned)maxU1) + (unsigned)uoff[i];  // uoff = min(u, maxU1-1) or a guard slot
    const float2 q = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(rec0) + (size_t)(off << 4));
    pb[i] = q.x;
    pl[i] = q.y;
  }
}

template <int K>
__device__ __forceinline__ void store_row(double* base, int row, unsigned U1s, int u0, const double (&v)[K]) {
  double* p = base + ((unsigned)row * U1s + (unsigned)u0);
  if constexpr (K == 1) {
    p[0] = v[0];
  } else {
#pragma unroll
    for (int i = 0; i < K; i += 2) *reinterpret_cast<double2*>(p + i) = make_double2(v[i], v[i + 1]);
  }
}

// log2(2^x + 2^y) with the sum in float64 and the correction term in float32
__device__ __forceinline__ double log2_add(double x, double y) {
  const float df = (float)(x - y);
  return fmax(x, y) + (double)fast_log2(1.0f + fast_exp2(-fabsf(df)));
}

template <int K>
__global__ __launch_bounds__(64) void k_sweep(Ws w, float* __restrict__ costs, const int32_t* __restrict__ in_len,
                                              const int32_t* __restrict__ lab_len, int B, int maxT, int maxU1,
                                              int b0) {
  constexpr int P = 4;  // diagonals of log-probs kept in flight ahead of the dependent chain
  const int b = b0 + (blockIdx.x >> 1);
  const int dir = blockIdx.x & 1;
  const int lane = threadIdx.x;
  const int Traw = in_len[b], Uraw = lab_len[b];
  const bool ok = (Traw >= 1) & (Traw <= maxT) & (Uraw >= 0) & (Uraw <= maxU1 - 1);
  if (!ok) {
    if (lane == 0) {
      atomicOr(&w.hdr->flags, RNNT_MI355X_FLAG_BAD_LENGTH);
      w.ll[dir * B + b] = 0.0;
      if (dir == 0) costs[b] = __builtin_nanf("");
    }
    return;
  }
  const int T = Traw, U = Uraw;
  const unsigned U1s = (unsigned)w.U1s;
  const size_t plane = (size_t)b * w.D * U1s;
  const float4* rec0 = w.rec + (size_t)b * (maxT + 2) * maxU1;  // frame -1 (guard), column 0
  const int u0 = lane * K;
  const int dlast = T - 1 + U;

  // columns beyond maxU1-1 have no record: they read the guard frame (frame index forced to -1 by a huge column)
  int ucol[K], uoff[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int u = u0 + i;
    const bool has = u < maxU1;
    ucol[i] = has ? u : (1 << 28);
    uoff[i] = has ? u : 0;
  }

  if (dir == 0) {
    double* alpha = w.alpha + plane;
    double a[K];
#pragma unroll
    for (int i = 0; i < K; ++i) a[i] = (u0 + i == 0) ? 0.0 : kNeg;
    store_row<K>(alpha, 0, U1s, u0, a);
    // one diagonal: a <- alpha(d) from alpha(d-1) and the log-probs of diagonal d-1
    auto step = [&](int d, const float (&pb)[K], const float (&pl)[K]) {
      double el[K];
#pragma unroll
      for (int i = 0; i < K; ++i) el[i] = a[i] + (double)pl[i];
      const double from_left = lane_shift_up(el[K - 1], kNeg);
#pragma unroll
      for (int i = 0; i < K; ++i) a[i] = log2_add(a[i] + (double)pb[i], (i == 0) ? from_left : el[i - 1]);
      store_row<K>(alpha, d, U1s, u0, a);
    };
    int d = 1;
    // remainder first, so that the pipelined loop below runs whole groups of P
    for (const int rem = dlast % P; d <= rem; ++d) {
      float pb[K], pl[K];
      load_lp_diag<K>(rec0, d - 1, maxT, maxU1, ucol, uoff, pb, pl);
      step(d, pb, pl);
    }
    if (d <= dlast) {
      float pb[P][K], pl[P][K];
#pragma unroll
      for (int p = 0; p < P; ++p) load_lp_diag<K>(rec0, d - 1 + p, maxT, maxU1, ucol, uoff, pb[p], pl[p]);
      for (; d <= dlast; d += P) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
          step(d + p, pb[p], pl[p]);
          load_lp_diag<K>(rec0, d + p - 1 + P, maxT, maxU1, ucol, uoff, pb[p], pl[p]);
          __builtin_amdgcn_sched_barrier(0);  // keep the refill P diagonals ahead of its use
        }
      }
    }
    // log2-likelihood = alpha(T-1,U) + log2 P(blank | T-1,U): column U of the last diagonal
#pragma unroll
    for (int i = 0; i < K; ++i) {
      if (u0 + i == U) {
        const double ll2 = a[i] + (double)rec0[(size_t)T * maxU1 + U].x;  // frame T-1 is at index T
        w.ll[b] = ll2;
        costs[b] = (float)(-ll2 * kLn2);
      }
    }
  } else {
    double* beta = w.beta + plane;
    // virtual diagonal dlast+1: beta(T,U) = log 1, everything else log 0
    double bt[K];
#pragma unroll
    for (int i = 0; i < K; ++i) bt[i] = (u0 + i == U) ? 0.0 : kNeg;
    // one diagonal: bt <- beta(d) from beta(d+1) and the log-probs of diagonal d
    auto step = [&](int d, const float (&pb)[K], const float (&pl)[K]) {
      const double from_right = lane_shift_down(bt[0], kNeg);
      double nb[K];
#pragma unroll
      for (int i = 0; i < K; ++i)
        nb[i] = log2_add(bt[i] + (double)pb[i], ((i == K - 1) ? from_right : bt[i + 1]) + (double)pl[i]);
#pragma unroll
      for (int i = 0; i < K; ++i) bt[i] = nb[i];
      store_row<K>(beta, d, U1s, u0, bt);
    };
    int d = dlast;
    for (int rem = (dlast + 1) % P; rem > 0; --rem, --d) {
      float pb[K], pl[K];
      load_lp_diag<K>(rec0, d, maxT, maxU1, ucol, uoff, pb, pl);
      step(d, pb, pl);
    }
    if (d >= 0) {
      float pb[P][K], pl[P][K];
#pragma unroll
      for (int p = 0; p < P; ++p) load_lp_diag<K>(rec0, d - p, maxT, maxU1, ucol, uoff, pb[p], pl[p]);
      for (; d >= 0; d -= P) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
          step(d - p, pb[p], pl[p]);
          load_lp_diag<K>(rec0, d - p - P, maxT, maxU1, ucol, uoff, pb[p], pl[p]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (lane == 0) w.ll[B + b] = bt[0];
  }
}

// ---------------------------------------------------------------------------
// k_grad: gradient w.r.t. the logits, written over them (or to `grad`).
//   dx[v] = s * ( 2^(x[v]*log2(e) + c2) - [v==blank] * pb - [v==label] * pl )
//   c2 = log2 gamma - lse,  pb = P(blank move | labels),  pl = P(label move | labels)
// Same block/tile decomposition and two-deep pipeline as k_lse.  A block first
// folds alpha, beta, the likelihood and the records of its cells into
// {c2, pb, pl, label} in LDS; rows that hold no lattice cell, or whose
// occupancy gamma is negligible, are marked inactive there and are zero-filled
// without being read.  Blocks are taken in the reverse of k_lse's order so
// that the rows k_lse read last (still in the memory-side cache) are re-read
// first.
// ---------------------------------------------------------------------------
template <int VEC, int NC, int G>
struct GradTile {
  float v[G][NC][VEC];
  float4 coef[G];  // {c2, pb, pl, label as int bits}; c2 = -inf marks an inactive row
  int rho[G];      // row inside the block, -1 = beyond the block
  int pass;
};

template <int VEC, int L, int NC, int G>
__global__ __launch_bounds__(kRowBlock) __attribute__((amdgpu_waves_per_eu(8))) void k_grad(const float* x, float* grad, const int32_t* __restrict__ labels,
                                                    const int32_t* __restrict__ in_len,
                                                    const int32_t* __restrict__ lab_len,
                                                    const float* __restrict__ grad_scale, RowGeom gm, Ws w) {
  constexpr int R = 64 / L;
  constexpr int S = kRowWaves * G * R;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane % L, r = lane / L;
  const int bid = gm.nblocks - 1 - (int)blockIdx.x;
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
  {
    const int U1s = w.U1s;
    const size_t plane = (size_t)b * w.D * U1s;
    const double ll2 = w.ll[b];
    for (int i = threadIdx.x; i < rows_blk; i += kRowBlock) {
      const int tt = i / maxU1;
      const int u = i - tt * maxU1;
      const int t = t0 + tt;
      float4 cf = make_float4(-INFINITY, 0.f, 0.f, __int_as_float(-1));
      if (t < T && u <= U) {
        const size_t ci = plane + (size_t)(t + u) * U1s + u;
        const double a_ll = w.alpha[ci] - ll2;
        const float4 rc = w.rec[((size_t)b * (gm.maxT + 2) + 1 + t) * maxU1 + u];  // {lpb, lpl, lse, -}
        const float g2 = (float)(a_ll + w.beta[ci]);   // log2 gamma
        if (g2 >= kPruneLog2) {
          float e_b = -INFINITY;  // 2^-inf = 0: no blank move out of this cell
          if (t < T - 1) e_b = (float)(a_ll + w.beta[ci + U1s]) + rc.x;
          else if (u == U) e_b = (float)a_ll + rc.x;
          float plv = 0.f;
          int lab = -1;
          if (u < U) {
            lab = labels[(size_t)b * (maxU1 - 1) + u];
            lab = (lab < 0 || lab >= V) ? 0 : lab;
            plv = fast_exp2((float)(a_ll + w.beta[ci + U1s + 1]) + rc.y);
          }
          cf = make_float4(g2 - rc.z, fast_exp2(e_b), plv, __int_as_float(lab));
        }
      }
      coef_s[i] = cf;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) w.hdr->n_partial = gm.nblocks_all;
  __syncthreads();

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
      if (!active) tl.coef[g].x = -INFINITY;
      const float* rp = xblk + (size_t)rho * V;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int c = j + L * (cur.pass * NC + i);
        if (active && c < nchunk) {
          load_vec<VEC>(rp + (size_t)c * VEC, tl.v[g][i]);
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
          } else {
#pragma unroll
            for (int q = 0; q < VEC; ++q) o[q] = 0.f;
          }
          store_vec<VEC>(gp + (size_t)c * VEC, o);
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

// ---------------------------------------------------------------------------
// Small-vocabulary variants (V <= 64: the phone-level vocabularies the
// reference trains on, V = 40 / 62).  A row is only V*4 <= 256 bytes, so
// splitting it over lanes spends more on cross-lane reductions than on the
// row itself.  Instead a wavefront streams tiles of 64 consecutive rows
// (16*V float4, coalesced) through registers into LDS and then every LANE
// owns one ROW: max, sum of exponentials and the gradient are plain serial
// loops over LDS with no cross-lane traffic at all, and the 64 records of a
// tile leave as one contiguous 1 KiB store.  The next tile's loads are in
// flight while the current one is processed.
// ---------------------------------------------------------------------------
constexpr int kSmallMaxV = 64;
constexpr int kSmallBlock = 128;                // two wavefronts, each with its own LDS tile
constexpr int kSmallWaves = kSmallBlock / 64;
constexpr int kSmallNld = kSmallMaxV * 64 / 4 / 64;  // float4 loads per lane per tile (max)

struct SmallGeom {
  int B, maxT, maxU1, V, blank;
  int frames;            // frames per utterance in memory = t_off + maxT
  int t_off;             // leading frames of every utterance that take no part in the lattice (a crop along T)
  int ntiles_full;       // tiles of 64 whole rows
  long long total_rows;  // B*frames*maxU1
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
template <int VEC>
__device__ __forceinline__ float small_row_lse2(const float* rowp, int V) {
  float m = -INFINITY;
  for (int i = 0; i < V; i += VEC) {
    float c[VEC];
    if constexpr (VEC == 4) { const float4 q = *reinterpret_cast<const float4*>(rowp + i); c[0] = q.x; c[1] = q.y; c[2] = q.z; c[3] = q.w; }
    else if constexpr (VEC == 2) { const float2 q = *reinterpret_cast<const float2*>(rowp + i); c[0] = q.x; c[1] = q.y; }
    else { c[0] = rowp[i]; }
#pragma unroll
    for (int q = 0; q < VEC; ++q) m = fmaxf(m, c[q]);
  }
  const float m2 = m * kLog2e;
  float s = 0.f;
  for (int i = 0; i < V; i += VEC) {
    float c[VEC];
    if constexpr (VEC == 4) { const float4 q = *reinterpret_cast<const float4*>(rowp + i); c[0] = q.x; c[1] = q.y; c[2] = q.z; c[3] = q.w; }
    else if constexpr (VEC == 2) { const float2 q = *reinterpret_cast<const float2*>(rowp + i); c[0] = q.x; c[1] = q.y; }
    else { c[0] = rowp[i]; }
#pragma unroll
    for (int q = 0; q < VEC; ++q) s += fast_exp2(__builtin_fmaf(c[q], kLog2e, -m2));
  }
  return m2 + fast_log2(s);
}

// Writes the record of one row (and the guard-frame entries next to frames 0 and maxT-1).
__device__ __forceinline__ void small_write_record(const Ws& w, const SmallGeom& g, int b, int t, int u, float4 rc) {
  float4* frame0 = w.rec + ((size_t)b * (g.maxT + 2) + 1) * g.maxU1;
  frame0[(size_t)t * g.maxU1 + u] = rc;
  const float4 neg = make_float4(kNegF, kNegF, 0.f, 0.f);
  if (t == 0) (frame0 - g.maxU1)[u] = neg;
  if (t == g.maxT - 1) (frame0 + (size_t)g.maxT * g.maxU1)[u] = neg;
}

template <int VEC>
__global__ __launch_bounds__(kSmallBlock) void k_lse_small(const float* __restrict__ x,
                                                           const int32_t* __restrict__ labels,
                                                           const int32_t* __restrict__ in_len,
                                                           const int32_t* __restrict__ lab_len, SmallGeom g, Ws w) {
  extern __shared__ float4 lds4[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int V = g.V;
  const int tile_f4 = 16 * V;  // 64 rows * V floats / 4
  float4* tile4 = lds4 + wv * tile_f4;
  const float* rowp = reinterpret_cast<const float*>(tile4) + lane * V;
  const int gw = blockIdx.x * kSmallWaves + wv, nw = gridDim.x * kSmallWaves;
  const f4v* x4 = reinterpret_cast<const f4v*>(x);

  int T = 0;
  auto row_valid = [&](int n, int& b, int& t, int& u, int& U, bool& in_range) {
    small_decode((long long)n * 64 + lane, g, in_len, lab_len, b, t, u, T, U, in_range);
    return in_range & (t < T) & (u <= U);
  };

  f4v st[kSmallNld];
  auto issue = [&](int n, bool any) {
    if (!any) return;
    const f4v* src = x4 + (size_t)n * tile_f4;
#pragma unroll
    for (int i = 0; i < kSmallNld; ++i) {
      const int q = lane + 64 * i;
      if (q < tile_f4) st[i] = __builtin_nontemporal_load(src + q);
    }
  };

  int n = gw;
  int b = -1, t = 0, u = 0, U = 0;
  bool in_range = false;
  bool valid = false, any = false;
  if (n < g.ntiles_full) {
    valid = row_valid(n, b, t, u, U, in_range);
    any = __ballot(valid) != 0ull;
    issue(n, any);
  }
  for (; n < g.ntiles_full; n += nw) {
    // stage the tile into LDS (only this wavefront touches it: in-order LDS, no barrier needed)
    if (any) {
#pragma unroll
      for (int i = 0; i < kSmallNld; ++i) {
        const int q = lane + 64 * i;
        if (q < tile_f4) tile4[q] = make_float4(st[i].x, st[i].y, st[i].z, st[i].w);
      }
    }
    const int cb = b, ct = t, cu = u, cU = U;
    const bool cvalid = valid, cany = any, cin = in_range;
    // next tile: decode and put its loads in flight before working on this one
    const int nn = n + nw;
    if (nn < g.ntiles_full) {
      valid = row_valid(nn, b, t, u, U, in_range);
      any = __ballot(valid) != 0ull;
      issue(nn, any);
    }
    float4 rc = make_float4(kNegF, kNegF, 0.f, 0.f);
    if (cany && cvalid) {
      const float lse2 = small_row_lse2<VEC>(rowp, V);
      float lpl = kNegF;
      if (cu < cU) {
        int lab = labels[(size_t)cb * (g.maxU1 - 1) + cu];
        if (lab < 0 || lab >= V) {
          atomicOr(&w.hdr->flags, RNNT_MI355X_FLAG_BAD_LABEL);
          lab = 0;
        }
        lpl = __builtin_fmaf(rowp[lab], kLog2e, -lse2);
      }
      rc = make_float4(__builtin_fmaf(rowp[g.blank], kLog2e, -lse2), lpl, lse2, 0.f);
    }
    if (cin) small_write_record(w, g, cb, ct, cu, rc);
  }
  // the last, partial tile (fewer than 64 rows): one wavefront reads its rows straight from global memory
  if (gw == 0 && (long long)g.ntiles_full * 64 < g.total_rows) {
    const long long row = (long long)g.ntiles_full * 64 + lane;
    b = -1;
    small_decode(row, g, in_len, lab_len, b, t, u, T, U, in_range);
    if (in_range) {
      float4 rc = make_float4(kNegF, kNegF, 0.f, 0.f);
      if (t < T && u <= U) {
        const float* rp = x + (size_t)row * V;
        const float lse2 = small_row_lse2<1>(rp, V);
        float lpl = kNegF;
        if (u < U) {
          int lab = labels[(size_t)b * (g.maxU1 - 1) + u];
          if (lab < 0 || lab >= V) {
            atomicOr(&w.hdr->flags, RNNT_MI355X_FLAG_BAD_LABEL);
            lab = 0;
          }
          lpl = __builtin_fmaf(rp[lab], kLog2e, -lse2);
        }
        rc = make_float4(__builtin_fmaf(rp[g.blank], kLog2e, -lse2), lpl, lse2, 0.f);
      }
      small_write_record(w, g, b, t, u, rc);
    }
  }
}

// Everything the gradient of a row needs about its lattice cell, as loaded (no arithmetic yet, so that the
// loads can be issued a tile ahead of their use).
struct CellRaw {
  double a, be, be_t, be_u, ll;
  float4 rc;  // {lpb, lpl, lse, -}
  int lab;
  int flags;  // 1 lattice cell, 2 t < T-1, 4 u == U, 8 u < U
};

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
    r.flags = 1 | (t < T - 1 ? 2 : 0) | (u == U ? 4 : 0) | (u < U ? 8 : 0);
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

// {c2, pb, pl, label} of the cell (c2 = -inf: the row's gradient is exactly zero)
__device__ __forceinline__ float4 cell_coef(const CellRaw& r) {
  float4 cf = make_float4(-INFINITY, 0.f, 0.f, __int_as_float(-1));
  if (r.flags & 1) {
    const double a_ll = r.a - r.ll;
    const float g2 = (float)(a_ll + r.be);  // log2 gamma
    if (g2 >= kPruneLog2) {
      float e_b = -INFINITY;  // 2^-inf = 0: no blank move out of this cell
      if (r.flags & 2) e_b = (float)(a_ll + r.be_t) + r.rc.x;
      else if (r.flags & 4) e_b = (float)a_ll + r.rc.x;
      const float plv = (r.flags & 8) ? fast_exp2((float)(a_ll + r.be_u) + r.rc.y) : 0.f;
      cf = make_float4(g2 - r.rc.z, fast_exp2(e_b), plv, __int_as_float(r.lab));
    }
  }
  return cf;
}

template <int VEC>
__device__ __forceinline__ float small_row_grad(const float* src, float* dst, int V, int blank, float4 cf, float sc) {
  const float c2 = cf.x, pb = cf.y, pl = cf.z;
  const int lab = __float_as_int(cf.w);
  float sq = 0.f;
  for (int i = 0; i < V; i += VEC) {
    float c[VEC];
    if constexpr (VEC == 4) { const float4 q = *reinterpret_cast<const float4*>(src + i); c[0] = q.x; c[1] = q.y; c[2] = q.z; c[3] = q.w; }
    else if constexpr (VEC == 2) { const float2 q = *reinterpret_cast<const float2*>(src + i); c[0] = q.x; c[1] = q.y; }
    else { c[0] = src[i]; }
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      float gv = fast_exp2(__builtin_fmaf(c[q], kLog2e, c2));
      gv -= (i + q == blank) ? pb : 0.f;
      gv -= (i + q == lab) ? pl : 0.f;
      gv *= sc;
      sq = __builtin_fmaf(gv, gv, sq);
      c[q] = gv;
    }
    if constexpr (VEC == 4) *reinterpret_cast<float4*>(dst + i) = make_float4(c[0], c[1], c[2], c[3]);
    else if constexpr (VEC == 2) *reinterpret_cast<float2*>(dst + i) = make_float2(c[0], c[1]);
    else dst[i] = c[0];
  }
  return sq;
}

template <int VEC>
__global__ __launch_bounds__(kSmallBlock) void k_grad_small(const float* x, float* grad,
                                                            const int32_t* __restrict__ labels,
                                                            const int32_t* __restrict__ in_len,
                                                            const int32_t* __restrict__ lab_len,
                                                            const float* __restrict__ grad_scale, SmallGeom g, Ws w) {
  extern __shared__ float4 lds4[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int V = g.V;
  const int tile_f4 = 16 * V;
  float4* tile4 = lds4 + wv * tile_f4;
  float* rowp = reinterpret_cast<float*>(tile4) + lane * V;
  const int gw = blockIdx.x * kSmallWaves + wv, nw = gridDim.x * kSmallWaves;
  const f4v* x4 = reinterpret_cast<const f4v*>(x);
  f4v* g4 = reinterpret_cast<f4v*>(grad);
  if (blockIdx.x == 0 && threadIdx.x == 0) w.hdr->n_partial = gridDim.x;

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

  f4v st[kSmallNld];
  auto issue = [&](int n, unsigned long long active) {
    const f4v* src = x4 + (size_t)n * tile_f4;
#pragma unroll
    for (int i = 0; i < kSmallNld; ++i) {
      const int q = lane + 64 * i;
      if (chunk_needed(active, i) && q < tile_f4) st[i] = __builtin_nontemporal_load(src + q);
    }
  };

  float sqsum = 0.f;
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
    cf = cell_coef(r0);
    active = __ballot(cf.x > -INFINITY);
    issue(last - k, active);
    if (k + nw < g.ntiles_full) raw = raw_of(last - (k + nw), sc_raw);
  }
  for (; k < g.ntiles_full; k += nw) {
    const int n = last - k;
#pragma unroll
    for (int i = 0; i < kSmallNld; ++i) {
      const int q = lane + 64 * i;
      if (chunk_needed(active, i) && q < tile_f4) tile4[q] = make_float4(st[i].x, st[i].y, st[i].z, st[i].w);
    }
    const float4 ccf = cf;
    const float csc = sc;
    if (k + nw < g.ntiles_full) {
      cf = cell_coef(raw);  // loaded one iteration ago
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
    for (int i = 0; i < kSmallNld; ++i) {
      const int q = lane + 64 * i;
      if (q < tile_f4) {
        const float4 o = tile4[q];
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
      const float4 c = cell_coef(cell_load(w, labels, b, t, u, T, U, g.maxT, g.maxU1, V, in_range));
      float* gp = grad + (size_t)row * V;
      if (c.x > -INFINITY) {
        sqsum += small_row_grad<1>(x + (size_t)row * V, gp, V, g.blank, c, grad_scale ? grad_scale[b] : 1.f);
      } else {
        for (int i = 0; i < V; ++i) gp[i] = 0.f;
      }
    }
  }
  __shared__ float red[kSmallWaves];
  sqsum = group_sum<64>(sqsum);
  if (lane == 0) red[wv] = sqsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kSmallWaves; ++i) s += red[i];
    w.partial[blockIdx.x] = s;
  }
}

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
//   k_joint_prep    mf2, mg2 (base-2 row maxima), F, G
//   k_joint_gather  blank / label parts of every cell's record
//   k_joint_lse_mm  S -> records {lpb, lpl, log2 S}          (then k_sweep as usual)
//   k_joint_coef    {c, pb, pl, label code} and C of every cell from alpha, beta, ll
//   k_joint_df_mm, k_joint_dg_mm   the two gradient contractions
//   k_joint_labelmap               chains of label positions per symbol: the blank / label terms of d f are
//                                  subtracted in the epilogue of its contraction
//   k_joint_corr_g                 the blank / label terms of d g
//   k_joint_fix_f, k_joint_fix_g   exponent-domain contributions of flagged cells (exit at once when there are none)
// ---------------------------------------------------------------------------
struct JointGeom {
  int B, maxT, maxU1, V, blank;
  size_t fstride;  // elements between utterances of f (>= maxT*V): lets f be a crop along T of a larger tensor
  size_t dfstride; // same for d f
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
                                                    JointGeom jg, Ws w) {
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
  if ((lane & 15) == 0 && row_ok) (isf ? w.mf2 + b * jg.maxT : w.mg2 + b * jg.maxU1)[row] = m;
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

// One thread per (b, t, u): the part of the cell's record that does not need the sum over V,
//   {f2(t,blank) + g2(u,blank), f2(t,y_{u+1}) + g2(u,y_{u+1}), 0, 0}   (f2 = f*log2e - mf2, g2 likewise)
// Entries that are no lattice cell get "log 0".
__global__ __launch_bounds__(256) void k_joint_gather(const float* __restrict__ f, const float* __restrict__ g,
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
  float4 rc = make_float4(kNegF, kNegF, 0.f, 0.f);
  if (t < T && u <= U) {
    const float* fr = f + (size_t)b * jg.fstride + (size_t)t * jg.V;
    const float* gr = g + ((size_t)b * jg.maxU1 + u) * jg.V;
    const float mf2 = w.mf2[b * jg.maxT + t], mg2 = w.mg2[b * jg.maxU1 + u];
    rc.x = __builtin_fmaf(fr[jg.blank], kLog2e, -mf2) + __builtin_fmaf(gr[jg.blank], kLog2e, -mg2);
    if (u < U) {
      int lab = labels[(size_t)b * (jg.maxU1 - 1) + u];
      if (lab < 0 || lab >= jg.V) {
        atomicOr(&w.hdr->flags, RNNT_MI355X_FLAG_BAD_LABEL);
        lab = 0;
      }
      rc.y = __builtin_fmaf(fr[lab], kLog2e, -mf2) + __builtin_fmaf(gr[lab], kLog2e, -mg2);
    }
  }
  w.rec[((size_t)b * (jg.maxT + 2) + 1 + t) * jg.maxU1 + u] = rc;
  const float4 neg = make_float4(kNegF, kNegF, 0.f, 0.f);
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
  // Records of the tile's cells: {x - ls, y - ls, ls, flag}.  All 16 records of an MFMA tile are loaded first
  // (lanes without a cell re-read a neighbouring one), then updated and stored, so that the memory round trips
  // overlap.  Cells whose sum underflowed are left for the slow loop below.
  float4* frame0 = w.rec + ((size_t)b * (jg.maxT + 2) + 1) * jg.maxU1;
  unsigned badmask = 0;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int u = u0 + 32 * j + l32;
    const int uc = min(u, U);
    float4 rc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rc[r] = frame0[(size_t)min(t0 + mm_row(r, lane), T - 1) * jg.maxU1 + uc];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = t0 + mm_row(r, lane);
      const bool valid = t < T && u <= U;
      const float S = acc[j][r];
      const bool bad = valid && !(S >= kJointMinSum);
      badmask |= bad ? (1u << (16 * j + r)) : 0u;
      if (valid && !bad) {
        const float ls = fast_log2(S);
        float4 o = rc[r];
        o.x -= ls;
        o.y = (u < U) ? o.y - ls : kNegF;
        o.z = ls;
        o.w = 0.f;
        frame0[(size_t)t * jg.maxU1 + u] = o;
      }
    }
  }
  if (__ballot(badmask != 0)) {  // rare: redo those cells with the exact maximum, one at a time, all lanes helping
    int n_bad = 0;
    for (int q = 0; q < 16 * NT; ++q) {
      const int j = q >> 4, r = q & 15;
      const int t = t0 + mm_row(r, lane), u = u0 + 32 * j + l32;
      const bool bad = (badmask >> q) & 1u;
      unsigned long long m = __ballot(bad);
      float ls = 0.f;
      while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const float e = joint_exact_ls(f, g, jg, w, b, __shfl(t, src), __shfl(u, src), lane);
        if (lane == src) ls = e;
      }
      if (bad) {
        float4* cell = frame0 + (size_t)t * jg.maxU1 + u;
        float4 o = *cell;
        o.x -= ls;
        o.y = (u < U) ? o.y - ls : kNegF;
        o.z = ls;
        o.w = 1.f;
        *cell = o;
        ++n_bad;
      }
    }
    if (n_bad) atomicAdd(&w.hdr->n_fallback, n_bad);
  }
}

// One block per (b, 8 frames, 32 label positions) of the padded lattice, one thread per cell: {c, s*pb, s*pl,
// label code} (c = log2 gamma - log2 S, -inf when the cell is pruned) and C = s * 2^c for the contractions (0 where
// pruned, flagged, or outside the lattice), written in both tiled orientations (the transposed one through LDS so
// that both leave as runs of 32 bytes).
__global__ __launch_bounds__(256) void k_joint_coef(const int32_t* __restrict__ labels,
                                                    const int32_t* __restrict__ in_len,
                                                    const int32_t* __restrict__ lab_len,
                                                    const float* __restrict__ grad_scale, JointGeom jg, Ws w) {
  __shared__ float tile[8][33];
  const int ub = w.U1p >> 5, tb = w.Tp >> 3;
  const int b = blockIdx.x / (ub * tb);
  const int r = blockIdx.x - b * ub * tb;
  const int t0 = (r / ub) * 8, u0 = (r - (r / ub) * ub) * 32;
  const int tt = threadIdx.x >> 5, uu = threadIdx.x & 31;
  const int t = t0 + tt, u = u0 + uu;
  float cm = 0.f, pbv = 0.f;
  if (t < jg.maxT && u < jg.maxU1) {
    int T, U;
    utt_dims(in_len, lab_len, b, jg.maxT, jg.maxU1, T, U);
    const CellRaw raw = cell_load(w, labels, b, t, u, T, U, jg.maxT, jg.maxU1, jg.V, true);
    float4 cf = cell_coef(raw);
    const float sc = grad_scale ? grad_scale[b] : 1.f;
    cf.y *= sc;
    cf.z *= sc;
    const bool live = cf.x > -INFINITY;
    const bool flagged = live && raw.rc.w != 0.f;
    cf.w = __int_as_float((raw.lab + 1) | (flagged ? kJointFlagBit : 0));
    if (live && !flagged) cm = sc * fast_exp2(cf.x);
    pbv = cf.y;
    w.coef[((size_t)b * jg.maxT + t) * jg.maxU1 + u] = cf;
  }
  // the 32 positions of a frame in this block are one half wavefront: their blank terms, summed
  const float pbsum = group_sum<32>(pbv);
  if (uu == 0) w.rowpart[((size_t)b * w.Tp + t) * ub + (u0 >> 5)] = pbsum;
  const size_t per_b = (size_t)w.Tp * w.U1p;
  w.Ctu[b * per_b + tiled_off(t, u, w.U1p >> 3)] = cm;
  tile[tt][uu] = cm;
  __syncthreads();
  const int u2 = threadIdx.x >> 3, t2 = threadIdx.x & 7;
  w.Cut[b * per_b + tiled_off(u0 + u2, t0 + t2, w.Tp >> 3)] = tile[t2][u2];
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

template <int NT, bool CORR>
__device__ __forceinline__ void joint_grad_mm(const float* __restrict__ a_base, int a_chunks, int kmax,
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
  // Two register sets used alternately: the loads of one are in flight during the 4*NT MFMAs of the other.
  f4v a0, a1;
  vecn bv0[4], bv1[4];
  auto load_set = [&](int k0, f4v& a, vecn (&bv)[4]) {
    a = ld4(ap + ((size_t)(k0 >> 3) << 8));
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const vecn*>(bp + b_off(k0, i));
  };
  auto mfma_set = [&](const f4v& a, const vecn (&bv)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], fvec_get<NT>(bv[i], j), acc[j], 0, 0, 0);
  };
  if (kmax > 0) {
    load_set(0, a0, bv0);
    for (int k0 = 0; k0 < kmax; k0 += 16) {
      const bool second = k0 + 8 < kmax;  // uniform
      load_set(second ? k0 + 8 : k0, a1, bv1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_set(a0, bv0);
      __builtin_amdgcn_sched_barrier(0);
      if (second) {
        load_set(min(k0 + 16, kmax - 8), a0, bv0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_set(a1, bv1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
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
          float corr = 0.f;
          for (int kk = k; kk >= 0; kk = dc.nxt[kk]) {
            if (kk == dc.blank_pos) {
              for (int q = 0; q < dc.nub; ++q) corr += dc.rowpart[(size_t)m * dc.nub + q];
            } else {
              corr += dc.coef[(size_t)m * dc.maxU1 + kk].z;
            }
          }
          acc[j][r] -= corr;
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

// d f(t,v) = F(t,v) * sum_u C(t,u) G(u,v): M = frames, N = vocabulary, K = label positions.
// CORR: the blank / label terms are subtracted in the epilogue (k_joint_labelmap has run); otherwise
// k_joint_corr_f follows.
template <int NT, bool CORR>
__global__ __launch_bounds__(256) void k_joint_df_mm(float* __restrict__ df, const int32_t* __restrict__ in_len,
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
  joint_grad_mm<NT, CORR>(w.Ctu + cb, w.U1p >> 3, kmax, w.Gp + (size_t)mt.b * w.U1p * w.Vp,
                          w.Fp + (size_t)mt.b * w.Tp * w.Vp, w.Vp, mt.m0, mt.n0, jg.maxT, jg.V,
                          df + (size_t)mt.b * jg.dfstride, vec_ok, threadIdx.x & 63, dc);
}

// d g(u,v) = G(u,v) * sum_t C(t,u) F(t,v): M = label positions, N = vocabulary, K = frames.
template <int NT>
__global__ __launch_bounds__(256) void k_joint_dg_mm(float* __restrict__ dg, const int32_t* __restrict__ in_len,
                                                     const int32_t* __restrict__ lab_len, JointGeom jg, Ws w,
                                                     bool vec_ok) {
  const MmTile mt = mm_tile(jg.B, w.U1p / 32, w.Vp / (32 * NT), 32 * NT);
  if (!mt.ok) return;
  int T, U;
  utt_dims(in_len, lab_len, mt.b, jg.maxT, jg.maxU1, T, U);
  const int kmax = (mt.m0 <= U) ? (T + 7) & ~7 : 0;  // C is zero in frames >= T_b and columns > U_b
  const size_t cb = (size_t)mt.b * w.Tp * w.U1p;
  joint_grad_mm<NT, false>(w.Cut + cb, w.Tp >> 3, kmax, w.Fp + (size_t)mt.b * w.Tp * w.Vp,
                           w.Gp + (size_t)mt.b * w.U1p * w.Vp, w.Vp, mt.m0, mt.n0, jg.maxU1, jg.V,
                           dg + (size_t)mt.b * jg.maxU1 * jg.V, vec_ok, threadIdx.x & 63, DfCorr{});
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
    float sacc = 0.f;
    for (int k = u; k >= 0; k = nxt[k]) sacc += vr[k];
    df[(size_t)b * jg.dfstride + (size_t)(t0 + i) * jg.V + tgt[u]] -= sacc;
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

// Blank and label terms of d g: 64 label positions per block, the frames split over the block's 16 waves.
__global__ __launch_bounds__(1024) void k_joint_corr_g(float* __restrict__ dg, const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ in_len,
                                                       const int32_t* __restrict__ lab_len, JointGeom jg, Ws w) {
  __shared__ float red[2][16][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ublocks = (jg.maxU1 + 63) / 64;
  const int b = blockIdx.x / ublocks;
  const int u = (blockIdx.x - b * ublocks) * 64 + lane;
  int T, U;
  utt_dims(in_len, lab_len, b, jg.maxT, jg.maxU1, T, U);
  float sb = 0.f, sl = 0.f;
  if (u <= U) {
    const float4* cfb = w.coef + (size_t)b * jg.maxT * jg.maxU1 + u;
    for (int t = wv; t < T; t += 16) {
      const float4 cf = cfb[(size_t)t * jg.maxU1];
      sb += cf.y;
      sl += cf.z;
    }
  }
  red[0][wv][lane] = sb;
  red[1][wv][lane] = sl;
  __syncthreads();
  if (wv == 0 && u <= U && T > 0) {
    sb = 0.f;
    sl = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      sb += red[0][i][lane];
      sl += red[1][i][lane];
    }
    float* out = dg + ((size_t)b * jg.maxU1 + u) * jg.V;
    out[jg.blank] -= sb;
    if (u < U) {
      const int l = labels[(size_t)b * (jg.maxU1 - 1) + u];
      out[(l < 0 || l >= jg.V) ? 0 : l] -= sl;
    }
  }
}

// Exponent-domain contribution of the flagged cells to d f: a wave owns TT frames and loops over u.
template <int KE, int TT>
__global__ __launch_bounds__(256) void k_joint_fix_f(const float* __restrict__ f, const float* __restrict__ g,
                                                     float* __restrict__ df, const int32_t* __restrict__ in_len,
                                                     const int32_t* __restrict__ lab_len,
                                                     const float* __restrict__ grad_scale, JointGeom jg, Ws w) {
  if (w.hdr->n_fallback == 0) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ntile = (jg.maxT + TT - 1) / TT;
  const int blocks_per_b = (ntile + 3) / 4;
  const int b = blockIdx.x / blocks_per_b;
  const int tile = (blockIdx.x - b * blocks_per_b) * 4 + wv;
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
__global__ __launch_bounds__(256) void k_joint_fix_g(const float* __restrict__ f, const float* __restrict__ g,
                                                     float* __restrict__ dg, const int32_t* __restrict__ in_len,
                                                     const int32_t* __restrict__ lab_len,
                                                     const float* __restrict__ grad_scale, JointGeom jg, Ws w) {
  if (w.hdr->n_fallback == 0) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ntile = (jg.maxU1 + UU - 1) / UU;
  const int blocks_per_b = (ntile + 3) / 4;
  const int b = blockIdx.x / blocks_per_b;
  const int tile = (blockIdx.x - b * blocks_per_b) * 4 + wv;
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

// zeroes the first n floats of every utterance (the frames a crop along T leaves out of the lattice)
__global__ __launch_bounds__(256) void k_zero_frames(float* __restrict__ g, size_t batch_stride, size_t n) {
  float* p = g + (size_t)blockIdx.y * batch_stride;
  const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (i0 + k < n) p[i0 + k] = 0.f;
}

__global__ __launch_bounds__(64) void k_init(Ws w, int n_partial) {
  if (threadIdx.x == 0) {
    w.hdr->flags = 0;
    w.hdr->n_fallback = 0;
    w.hdr->n_partial = n_partial;
    w.hdr->partial_off = w.partial_off;
  }
}

// out2[0] = sum costs, out2[1] = sum of the per-block partials of the last k_grad.
__global__ __launch_bounds__(1024) void k_reduce_scalars(const float* __restrict__ costs, int B, const char* ws_base,
                                                         double* __restrict__ out2) {
  __shared__ double sh[2][1024];
  const Header* hdr = reinterpret_cast<const Header*>(ws_base);
  const float* partial = reinterpret_cast<const float*>(ws_base + hdr->partial_off);
  const int n = hdr->n_partial;
  double c = 0.0, s = 0.0;
  for (int i = threadIdx.x; i < B; i += 1024) c += (double)costs[i];
  for (int i = threadIdx.x; i < n; i += 1024) s += (double)partial[i];
  sh[0][threadIdx.x] = c;
  sh[1][threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out2[0] = sh[0][0];
    out2[1] = sh[1][0];
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

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
  const int cpl = (nchunk + c.L - 1) / c.L;  // chunks per lane per row
  c.nc = cpl >= 3 ? 4 : cpl;
  c.g = 4 / c.nc;
  const int rows_per_iter = kRowWaves * c.g * (64 / c.L);
  // frames per block: about 24 tiles per wave, so that the pipeline fill/drain is amortised
  int TT = (24 * rows_per_iter + maxU1 - 1) / maxU1;
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
    else if ((cfg).L == 32) KERNEL<VEC, 32, 1, 4><<<grid_, blk_, LDS, st>>>(__VA_ARGS__);                  \
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
  // the sweeps address an utterance's records with 32-bit byte offsets and 24-bit frame indices
  if (((long long)maxT + 2) * maxU1 >= (1LL << 28) || maxT >= (1 << 24) - 2) return RNNT_MI355X_UNSUPPORTED_SIZE;
  return RNNT_MI355X_OK;
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
  Ws w;
};

RowCfg sub_cfg(const Call& c, const void* p1, int b0, int nb) {
  // the widest vector access must also survive the batch strides
  const uintptr_t align_bits = reinterpret_cast<uintptr_t>(p1) | (uintptr_t)(c.xstride * 4) | (uintptr_t)(c.gstride * 4);
  RowCfg cfg = pick_row_cfg(c.B, c.maxT, c.maxU1, c.V, c.blank, c.logits, reinterpret_cast<const void*>(align_bits));
  cfg.gm.xstride = c.xstride;
  cfg.gm.gstride = c.gstride;
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
  g.ntiles_full = (int)(g.total_rows / 64);
  long long want = (g.ntiles_full + kSmallWaves - 1) / kSmallWaves;
  // persistent grid, exactly one resident round: tiles are dealt round-robin to the waves.  Residency per CU is
  // limited by LDS (160 KiB / tile bytes per block) and by registers (6 blocks of two waves).
  const long long lds_bytes = 16LL * 16 * c.V * kSmallWaves;
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

void launch_lse(const Call& c, int b0, int nb, hipStream_t st) {
  if (b0 == 0 && nb == c.B && use_small(c, c.logits)) {
    int blocks;
    const SmallGeom g = small_geom(c, &blocks);
    const size_t lds = sizeof(float4) * 16 * (size_t)c.V * kSmallWaves;
    const float* x0 = c.logits - (size_t)c.t_off * c.maxU1 * c.V;  // frame 0 of the uncropped tensor
    if (c.V % 4 == 0) k_lse_small<4><<<blocks, kSmallBlock, lds, st>>>(x0, c.labels, c.in_len, c.lab_len, g, c.w);
    else if (c.V % 2 == 0) k_lse_small<2><<<blocks, kSmallBlock, lds, st>>>(x0, c.labels, c.in_len, c.lab_len, g, c.w);
    else k_lse_small<1><<<blocks, kSmallBlock, lds, st>>>(x0, c.labels, c.in_len, c.lab_len, g, c.w);
    return;
  }
  const RowCfg cfg = sub_cfg(c, c.logits, b0, nb);
  RNNT_DISPATCH(k_lse, cfg, sizeof(float4) * (size_t)cfg.gm.TT * c.maxU1, c.logits, c.labels, c.in_len, c.lab_len,
                cfg.gm, c.w);
}

void launch_sweep(const Call& c, int b0, int nb, hipStream_t st) {
  const int grid = 2 * nb;
  switch (sweep_cols_per_lane(c.maxU1)) {
    case 1: k_sweep<1><<<grid, 64, 0, st>>>(c.w, c.costs, c.in_len, c.lab_len, c.B, c.maxT, c.maxU1, b0); break;
    case 2: k_sweep<2><<<grid, 64, 0, st>>>(c.w, c.costs, c.in_len, c.lab_len, c.B, c.maxT, c.maxU1, b0); break;
    case 4: k_sweep<4><<<grid, 64, 0, st>>>(c.w, c.costs, c.in_len, c.lab_len, c.B, c.maxT, c.maxU1, b0); break;
    case 8: k_sweep<8><<<grid, 64, 0, st>>>(c.w, c.costs, c.in_len, c.lab_len, c.B, c.maxT, c.maxU1, b0); break;
    default: k_sweep<16><<<grid, 64, 0, st>>>(c.w, c.costs, c.in_len, c.lab_len, c.B, c.maxT, c.maxU1, b0); break;
  }
}

void launch_grad(const Call& c, int b0, int nb, hipStream_t st) {
  if (b0 == 0 && nb == c.B && use_small(c, c.grad)) {
    int blocks;
    const SmallGeom g = small_geom(c, &blocks);
    const size_t lds = sizeof(float4) * 16 * (size_t)c.V * kSmallWaves;
    const size_t skip = (size_t)c.t_off * c.maxU1 * c.V;  // the cropped frames are rows without a cell: zero-filled
    const float* x0 = c.logits - skip;
    float* g0 = c.grad - skip;
    if (c.V % 4 == 0) k_grad_small<4><<<blocks, kSmallBlock, lds, st>>>(x0, g0, c.labels, c.in_len, c.lab_len, c.grad_scale, g, c.w);
    else if (c.V % 2 == 0) k_grad_small<2><<<blocks, kSmallBlock, lds, st>>>(x0, g0, c.labels, c.in_len, c.lab_len, c.grad_scale, g, c.w);
    else k_grad_small<1><<<blocks, kSmallBlock, lds, st>>>(x0, g0, c.labels, c.in_len, c.lab_len, c.grad_scale, g, c.w);
    return;
  }
  if (c.t_off > 0 && b0 == 0) {  // gradient of the cropped frames
    const size_t n = (size_t)c.t_off * c.maxU1 * c.V;
    const unsigned per_b = (unsigned)((n + 1023) / 1024);
    k_zero_frames<<<dim3(per_b, (unsigned)c.B), 256, 0, st>>>(c.grad - n, c.gstride, n);
  }
  const RowCfg cfg = sub_cfg(c, c.grad, b0, nb);
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

// forward + backward on the caller's stream.  ev_begin / ev_end (optional hipEvent_t) bracket the gradient
// launch.  (Running the latency-bound sweep of one half of the batch on a second stream underneath the
// streaming kernels of the other half was measured and rejected: the sweep is hidden, but two extra kernel
// tails and its interference cost as much as the 0.1 ms it saves.)
int launch_loss_grad(const Call& c, hipStream_t st, hipEvent_t ev_begin, hipEvent_t ev_end) {
  int rc = launch_forward(c, st);
  if (rc != RNNT_MI355X_OK) return rc;
  if (ev_begin && hipEventRecord(ev_begin, st) != hipSuccess) return RNNT_MI355X_LAUNCH_FAILURE;
  rc = launch_backward(c, st);
  if (ev_end && hipEventRecord(ev_end, st) != hipSuccess) return RNNT_MI355X_LAUNCH_FAILURE;
  return rc;
}

int make_call(Call* c, const float* logits, float* grad, const int32_t* labels, const int32_t* in_len,
              const int32_t* lab_len, int B, int maxT, int maxU1, int V, int blank, const float* grad_scale,
              float* costs, void* workspace, bool need_grad, bool need_costs, size_t xstride = 0, size_t gstride = 0,
              int t_off = 0) {
  const int rc = check_shape(B, maxT, maxU1, V, blank);
  if (rc != RNNT_MI355X_OK) return rc;
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
  k_init<<<1, 64, 0, st>>>(w, 0);
  const int groups = jg.B * ((w.Tp + w.U1p) / 4);
  const bool rows16 = jg.V % 4 == 0 && jg.fstride % 4 == 0 && reinterpret_cast<uintptr_t>(c.f) % 16 == 0 &&
                      reinterpret_cast<uintptr_t>(c.g) % 16 == 0;
  const unsigned pgrid = (unsigned)((groups + 3) / 4);
#define JOINT_PREP(NIT)                                                            \
  do {                                                                             \
    if (rows16) k_joint_prep<true, NIT><<<pgrid, 256, 0, st>>>(c.f, c.g, jg, w);   \
    else k_joint_prep<false, NIT><<<pgrid, 256, 0, st>>>(c.f, c.g, jg, w);         \
  } while (0)
  if (w.Vp <= 256) JOINT_PREP(4);
  else if (w.Vp <= 1024) JOINT_PREP(16);
  else if (w.Vp <= 2048) JOINT_PREP(32);
  else JOINT_PREP(0);
#undef JOINT_PREP
  const long long ncell = (long long)jg.B * jg.maxT * jg.maxU1;
  k_joint_gather<<<(unsigned)((ncell + 255) / 256), 256, 0, st>>>(c.f, c.g, c.labels, c.in_len, c.lab_len, jg, w);
  const int mt = w.Tp / 32;
  if (w.U1p <= 32)
    k_joint_lse_mm<1><<<mm_blocks(jg.B, mt, 1), 256, 0, st>>>(c.f, c.g, c.in_len, c.lab_len, jg, w);
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
    k_joint_dg_mm<1><<<mm_blocks(jg.B, mu, ns), 256, 0, st>>>(dg, c.in_len, c.lab_len, jg, w, vg);
  } else if (nt == 2) {
    k_joint_df_mm<2, false><<<mm_blocks(jg.B, mt, ns), 256, 0, st>>>(df, c.in_len, c.lab_len, jg, w, vf);
    k_joint_dg_mm<2><<<mm_blocks(jg.B, mu, ns), 256, 0, st>>>(dg, c.in_len, c.lab_len, jg, w, vg);
  } else {
    if (fold) k_joint_df_mm<4, true><<<mm_blocks(jg.B, mt, ns), 256, 0, st>>>(df, c.in_len, c.lab_len, jg, w, vf);
    else k_joint_df_mm<4, false><<<mm_blocks(jg.B, mt, ns), 256, 0, st>>>(df, c.in_len, c.lab_len, jg, w, vf);
    k_joint_dg_mm<4><<<mm_blocks(jg.B, mu, ns), 256, 0, st>>>(dg, c.in_len, c.lab_len, jg, w, vg);
  }
  if (!fold) {
    const int fb = jg.maxU1 <= 128 ? 32 : jg.maxU1 <= 256 ? 16 : 8;  // frames per block: (3 + fb) * (maxU1 + 1) words of LDS
    const size_t corr_lds = sizeof(int) * (3 + fb) * (size_t)(jg.maxU1 + 1);
    k_joint_corr_f<<<jg.B * ((jg.maxT + fb - 1) / fb), 256, corr_lds, st>>>(df, c.labels, c.in_len, c.lab_len, jg, w, fb);
  }
  k_joint_corr_g<<<jg.B * ((jg.maxU1 + 63) / 64), 1024, 0, st>>>(dg, c.labels, c.in_len, c.lab_len, jg, w);
#define JGRID_T(TT) dim3((unsigned)(jg.B * ((((jg.maxT + (TT)-1) / (TT)) + 3) / 4)))
#define JGRID_U(UU) dim3((unsigned)(jg.B * ((((jg.maxU1 + (UU)-1) / (UU)) + 3) / 4)))
  RNNT_JOINT_DISPATCH(k_joint_fix_f, ke, JGRID_T, 2, 1, 1, c.f, c.g, df, c.in_len, c.lab_len, grad_scale, jg, w);
  RNNT_JOINT_DISPATCH(k_joint_fix_g, ke, JGRID_U, 2, 1, 1, c.f, c.g, dg, c.in_len, c.lab_len, grad_scale, jg, w);
  return launch_status();
}

}  // namespace

extern "C" {

int rnnt_mi355x_version(void) { return 101; }

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
                           workspace, false, true);
  if (rc != RNNT_MI355X_OK) return rc;
  return launch_forward(c, static_cast<hipStream_t>(hip_stream));
}

int rnnt_mi355x_backward(const float* logits, float* grad, const int32_t* labels, const int32_t* in_len,
                         const int32_t* lab_len, int B, int maxT, int maxU1, int V, int blank, const float* grad_scale,
                         void* workspace, void* hip_stream) {
  Call c;
  const int rc = make_call(&c, logits, grad, labels, in_len, lab_len, B, maxT, maxU1, V, blank, grad_scale, nullptr,
                           workspace, true, false);
  if (rc != RNNT_MI355X_OK) return rc;
  return launch_backward(c, static_cast<hipStream_t>(hip_stream));
}

int rnnt_mi355x_forward_strided(const float* logits, size_t logits_batch_stride, const int32_t* labels,
                                const int32_t* in_len, const int32_t* lab_len, int B, int maxT, int maxU1, int V,
                                int blank, float* costs, void* workspace, void* hip_stream) {
  Call c;
  const int rc = make_call(&c, logits, nullptr, labels, in_len, lab_len, B, maxT, maxU1, V, blank, nullptr, costs,
                           workspace, false, true, logits_batch_stride, 0);
  if (rc != RNNT_MI355X_OK) return rc;
  return launch_forward(c, static_cast<hipStream_t>(hip_stream));
}

int rnnt_mi355x_backward_strided(const float* logits, size_t logits_batch_stride, float* grad,
                                 size_t grad_batch_stride, const int32_t* labels, const int32_t* in_len,
                                 const int32_t* lab_len, int B, int maxT, int maxU1, int V, int blank,
                                 const float* grad_scale, void* workspace, void* hip_stream) {
  Call c;
  const int rc = make_call(&c, logits, grad, labels, in_len, lab_len, B, maxT, maxU1, V, blank, grad_scale, nullptr,
                           workspace, true, false, logits_batch_stride, grad_batch_stride);
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
                           maxU1, V, blank, nullptr, costs, workspace, false, true, stride, stride, t_offset);
  if (rc != RNNT_MI355X_OK) return rc;
  return launch_forward(c, static_cast<hipStream_t>(hip_stream));
}

int rnnt_mi355x_backward_cropped(const float* logits_full, float* grad_full, const int32_t* labels,
                                 const int32_t* in_len, const int32_t* lab_len, int B, int T0, int t_offset,
                                 int maxU1, int V, int blank, const float* grad_scale, void* workspace,
                                 void* hip_stream) {
  if (t_offset < 0 || t_offset >= T0 || !logits_full || !grad_full) return RNNT_MI355X_INVALID_VALUE;
  if ((long long)B * T0 * maxU1 >= (1LL << 31)) return RNNT_MI355X_UNSUPPORTED_SIZE;
  const size_t frame = (size_t)maxU1 * V, stride = (size_t)T0 * frame;
  Call c;
  const int rc = make_call(&c, logits_full + t_offset * frame, grad_full + t_offset * frame, labels, in_len, lab_len,
                           B, T0 - t_offset, maxU1, V, blank, grad_scale, nullptr, workspace, true, false, stride,
                           stride, t_offset);
  if (rc != RNNT_MI355X_OK) return rc;
  return launch_backward(c, static_cast<hipStream_t>(hip_stream));
}

int rnnt_mi355x_backward_inplace(float* logits_inout, const int32_t* labels, const int32_t* in_len,
                                 const int32_t* lab_len, int B, int maxT, int maxU1, int V, int blank,
                                 const float* grad_scale, void* workspace, void* hip_stream) {
  return rnnt_mi355x_backward(logits_inout, logits_inout, labels, in_len, lab_len, B, maxT, maxU1, V, blank,
                              grad_scale, workspace, hip_stream);
}

int rnnt_mi355x_loss_grad(float* logits_inout, const int32_t* labels, const int32_t* in_len, const int32_t* lab_len,
                          int B, int maxT, int maxU1, int V, int blank, const float* grad_scale, float* costs,
                          void* workspace, void* hip_stream) {
  return rnnt_mi355x_loss_grad_events(logits_inout, labels, in_len, lab_len, B, maxT, maxU1, V, blank, grad_scale,
                                      costs, workspace, hip_stream, nullptr, nullptr);
}

int rnnt_mi355x_loss_grad_events(float* logits_inout, const int32_t* labels, const int32_t* in_len,
                                 const int32_t* lab_len, int B, int maxT, int maxU1, int V, int blank,
                                 const float* grad_scale, float* costs, void* workspace, void* hip_stream,
                                 void* ev_grad_begin, void* ev_grad_end) {
  Call c;
  const int rc = make_call(&c, logits_inout, logits_inout, labels, in_len, lab_len, B, maxT, maxU1, V, blank,
                           grad_scale, costs, workspace, true, true);
  if (rc != RNNT_MI355X_OK) return rc;
  return launch_loss_grad(c, static_cast<hipStream_t>(hip_stream), static_cast<hipEvent_t>(ev_grad_begin),
                          static_cast<hipEvent_t>(ev_grad_end));
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
                               void* workspace, void* hip_stream) {
  JointCall c;
  const int rc = make_joint_call(&c, f, f_batch_stride, g, labels, in_len, lab_len, B, maxT, maxU1, V, blank,
                                 df_batch_stride, workspace);
  if (rc != RNNT_MI355X_OK) return rc;
  if (!df || !dg) return RNNT_MI355X_INVALID_VALUE;
  return launch_joint_backward(c, grad_scale, df, dg, static_cast<hipStream_t>(hip_stream));
}

int rnnt_mi355x_reduce_scalars(const float* costs, int B, void* workspace, double* out2, void* hip_stream) {
  if (!costs || !workspace || !out2 || B < 1) return RNNT_MI355X_INVALID_VALUE;
  k_reduce_scalars<<<1, 1024, 0, static_cast<hipStream_t>(hip_stream)>>>(costs, B, static_cast<const char*>(workspace),
                                                                        out2);
  return launch_status();
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
