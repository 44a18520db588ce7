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
constexpr double kNeg = -1.0e30;   // "log 0" that stays finite under addition
constexpr int kMaxU1 = 512;
constexpr int kRowBlock = 256;     // threads per block of the streaming kernels
constexpr int kMaxRowBlocks = 2048;  // 256 CUs x 8 blocks
constexpr int kStatusInts = 64;

struct Ws {
  int32_t* status;   // [kStatusInts]
  double* ll;        // [2][B]  log2-likelihood from alpha / from beta
  float* partial;    // [kMaxRowBlocks] per-block sum of squared gradients
  float* lse;        // [B][D][U1s] log2-sum-exp2 of the row
  float* lpb;        // [B][D][U1s] log2 P(blank | t,u)
  float* lpl;        // [B][D][U1s] log2 P(label u+1 | t,u)
  double* alpha;     // [B][D][U1s]
  double* beta;      // [B][D][U1s]
  int D;             // diagonals = maxT + maxU1 - 1
  int U1s;           // row stride (maxU1 rounded up to 8)
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Lays the workspace out; returns total bytes.  base may be null (size query).
size_t ws_layout(void* base, int B, int maxT, int maxU1, Ws* w) {
  const int D = maxT + maxU1 - 1;
  const int U1s = (maxU1 + 7) / 8 * 8;
  const size_t plane = (size_t)B * D * U1s;
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  const size_t o_status = take(sizeof(int32_t) * kStatusInts);
  const size_t o_partial = take(sizeof(float) * kMaxRowBlocks);
  const size_t o_ll = take(sizeof(double) * 2 * B);
  const size_t o_lse = take(sizeof(float) * plane);
  const size_t o_lpb = take(sizeof(float) * plane);
  const size_t o_lpl = take(sizeof(float) * plane);
  const size_t o_alpha = take(sizeof(double) * plane);
  const size_t o_beta = take(sizeof(double) * plane);
  if (w) {
    w->status = reinterpret_cast<int32_t*>(p + o_status);
    w->ll = reinterpret_cast<double*>(p + o_ll);
    w->partial = reinterpret_cast<float*>(p + o_partial);
    w->lse = reinterpret_cast<float*>(p + o_lse);
    w->lpb = reinterpret_cast<float*>(p + o_lpb);
    w->lpl = reinterpret_cast<float*>(p + o_lpl);
    w->alpha = reinterpret_cast<double*>(p + o_alpha);
    w->beta = reinterpret_cast<double*>(p + o_beta);
    w->D = D;
    w->U1s = U1s;
  }
  return off;
}

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

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  } else if constexpr (VEC == 2) {
    const float2 q = *reinterpret_cast<const float2*>(p);
    v[0] = q.x; v[1] = q.y;
  } else {
    v[0] = *p;
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  } else {
    *p = v[0];
  }
}

template <int L>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = L / 2; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
template <int L>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = L / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Position of a logits row in the lattice.
struct RowPos {
  int b, t, u;
  int T, U;
  bool in_range;  // row < B*maxT*maxU1
  bool valid;     // a real lattice cell of utterance b
};

__device__ __forceinline__ RowPos decode_row(long long row, long long total_rows, int maxT, int maxU1,
                                             const int32_t* in_len, const int32_t* lab_len) {
  RowPos p;
  p.in_range = row < total_rows;
  const unsigned r = p.in_range ? (unsigned)row : 0u;
  const unsigned per_b = (unsigned)maxT * (unsigned)maxU1;
  p.b = (int)(r / per_b);
  const unsigned rem = r - (unsigned)p.b * per_b;
  p.t = (int)(rem / (unsigned)maxU1);
  p.u = (int)(rem - (unsigned)p.t * (unsigned)maxU1);
  utt_dims(in_len, lab_len, p.b, maxT, maxU1, p.T, p.U);
  p.valid = p.in_range && (p.t < p.T) && (p.u <= p.U);
  return p;
}

// ---------------------------------------------------------------------------
// k_lse: row log-sum-exp + blank/label log-probs.
//
// A wavefront is split into R = 64/L groups of L lanes; each group owns one
// row, lane j of the group owns the VEC-wide chunks j, j+L, j+2L, ...  One
// wave iteration covers G*R consecutive rows with G*NC vector loads in flight
// per lane (4 x 16 B for V % 4 == 0).
// ---------------------------------------------------------------------------
template <int VEC, int L, int NC, int G>
__global__ __launch_bounds__(kRowBlock) void k_lse(const float* __restrict__ x, const int32_t* __restrict__ labels,
                                                   const int32_t* __restrict__ in_len,
                                                   const int32_t* __restrict__ lab_len, int B, int maxT, int maxU1,
                                                   int V, int blank, Ws w, long long nit) {
  constexpr int R = 64 / L;
  const int lane = threadIdx.x & 63;
  const int j = lane % L;
  const int r = lane / L;
  const long long wave = (long long)blockIdx.x * (kRowBlock / 64) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (kRowBlock / 64);
  const int nchunk = V / VEC;
  const int cpl = (nchunk + L - 1) / L;  // chunks per lane per row
  const long long total_rows = (long long)B * maxT * maxU1;

  for (long long it = wave; it < nit; it += nwaves) {
    RowPos pos[G];
    const float* rowp[G];
    float extra[G];  // lane j==0: logit of blank; lane j==1: logit of the next label
    float m_run[G], s_run[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const long long row = (it * G + g) * R + r;
      pos[g] = decode_row(row, total_rows, maxT, maxU1, in_len, lab_len);
      rowp[g] = x + (size_t)(pos[g].in_range ? row : 0) * (size_t)V;
      m_run[g] = -INFINITY;
      s_run[g] = 0.f;
      extra[g] = 0.f;
      if (pos[g].valid) {
        if (j == 0) {
          extra[g] = rowp[g][blank];
        } else if (j == 1 && pos[g].u < pos[g].U) {
          int lab = labels[(size_t)pos[g].b * (maxU1 - 1) + pos[g].u];
          if (lab < 0 || lab >= V) {
            atomicOr(w.status, RNNT_MI355X_FLAG_BAD_LABEL);
            lab = 0;
          }
          extra[g] = rowp[g][lab];
        }
      }
    }
    for (int k0 = 0; k0 < cpl; k0 += NC) {
      float v[G][NC][VEC];
#pragma unroll
      for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
          const int c = j + L * (k0 + i);
          if (pos[g].valid && c < nchunk) {
            load_vec<VEC>(rowp[g] + (size_t)c * VEC, v[g][i]);
          } else {
#pragma unroll
            for (int q = 0; q < VEC; ++q) v[g][i][q] = -INFINITY;
          }
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
          for (int q = 0; q < VEC; ++q) mx = fmaxf(mx, v[g][i][q]);
        mx = group_max<L>(mx);
        // rows with no valid cell keep m = 0 so that nothing below is NaN
        const float m_new = pos[g].valid ? fmaxf(m_run[g], mx) : 0.f;
        const float m2 = m_new * kLog2e;
        float s = s_run[g] * fast_exp2(m_run[g] * kLog2e - m2);
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
          for (int q = 0; q < VEC; ++q) s += fast_exp2(__builtin_fmaf(v[g][i][q], kLog2e, -m2));
        m_run[g] = m_new;
        s_run[g] = s;
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float s_all = group_sum<L>(s_run[g]);
      const float lse2 = m_run[g] * kLog2e + fast_log2(s_all);
      if (pos[g].valid && j < 2) {
        const size_t ci = ((size_t)pos[g].b * w.D + (pos[g].t + pos[g].u)) * w.U1s + pos[g].u;
        const float lp2 = __builtin_fmaf(extra[g], kLog2e, -lse2);
        if (j == 0) {
          w.lse[ci] = lse2;
          w.lpb[ci] = lp2;
        } else if (pos[g].u < pos[g].U) {
          w.lpl[ci] = lp2;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// k_sweep: alpha (even blocks) and beta (odd blocks) anti-diagonal sweeps.
// One wavefront per (utterance, direction); lane l owns the K lattice columns
// u = l*K .. l*K+K-1, so one cross-lane shift per diagonal is the only
// communication and no barrier or LDS traffic sits on the dependent chain.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double log2_add(double p, double q) {
  const double mx = fmax(p, q);
  const double mn = fmin(p, q);
  const float df = (float)(mn - mx);  // <= 0
  return mx + (double)fast_log2(1.0f + fast_exp2(df));
}

template <int K>
__device__ __forceinline__ void load_lp_row(const float* base, int row, int U1s, int u0, float (&out)[K]) {
  if (u0 < U1s) {
    const float* p = base + (size_t)row * U1s + u0;
    if constexpr (K == 1) {
      out[0] = p[0];
    } else if constexpr (K == 2) {
      const float2 q = *reinterpret_cast<const float2*>(p);
      out[0] = q.x; out[1] = q.y;
    } else {
#pragma unroll
      for (int i = 0; i < K; i += 4) {
        const float4 q = *reinterpret_cast<const float4*>(p + i);
        out[i] = q.x; out[i + 1] = q.y; out[i + 2] = q.z; out[i + 3] = q.w;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < K; ++i) out[i] = 0.f;
  }
}

template <int K>
__device__ __forceinline__ void store_row(double* base, int row, int U1s, int u0, const double (&v)[K]) {
  if (u0 < U1s) {
    double* p = base + (size_t)row * U1s + u0;
    if constexpr (K == 1) {
      p[0] = v[0];
    } else {
#pragma unroll
      for (int i = 0; i < K; i += 2) *reinterpret_cast<double2*>(p + i) = make_double2(v[i], v[i + 1]);
    }
  }
}

template <int K>
__global__ __launch_bounds__(64) void k_sweep(Ws w, float* __restrict__ costs, const int32_t* __restrict__ in_len,
                                              const int32_t* __restrict__ lab_len, int B, int maxT, int maxU1) {
  constexpr int P = 4;  // diagonals of log-probs prefetched ahead of the dependent chain
  const int b = blockIdx.x >> 1;
  const int dir = blockIdx.x & 1;
  const int lane = threadIdx.x;
  const int Traw = in_len[b], Uraw = lab_len[b];
  const bool ok = (Traw >= 1) & (Traw <= maxT) & (Uraw >= 0) & (Uraw <= maxU1 - 1);
  if (!ok) {
    if (lane == 0) {
      atomicOr(w.status, RNNT_MI355X_FLAG_BAD_LENGTH);
      w.ll[dir * B + b] = 0.0;
      if (dir == 0) costs[b] = __builtin_nanf("");
    }
    return;
  }
  const int T = Traw, U = Uraw;
  const int U1s = w.U1s;
  const size_t plane = (size_t)b * w.D * U1s;
  const float* lpb = w.lpb + plane;
  const float* lpl = w.lpl + plane;
  const int u0 = lane * K;
  const int dlast = T - 1 + U;

  // per-column validity window: cell (d-u, u) exists iff u <= U and lo <= d <= hi
  int lo[K], hi[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int u = u0 + i;
    lo[i] = u;
    hi[i] = (u <= U) ? u + T - 1 : -1;  // empty window for columns beyond U
  }

  float pb[P][K], pl[P][K];

  if (dir == 0) {
    double* alpha = w.alpha + plane;
    double a[K];
#pragma unroll
    for (int i = 0; i < K; ++i) a[i] = (u0 + i == 0) ? 0.0 : kNeg;
    store_row<K>(alpha, 0, U1s, u0, a);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int row = min(p, dlast);
      load_lp_row<K>(lpb, row, U1s, u0, pb[p]);
      load_lp_row<K>(lpl, row, U1s, u0, pl[p]);
    }
    for (int d0 = 1; d0 <= dlast; d0 += P) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int d = d0 + p;
        if (d <= dlast) {
          double eb[K], el[K];
#pragma unroll
          for (int i = 0; i < K; ++i) {
            const bool src_ok = (d - 1 >= lo[i]) & (d - 1 <= hi[i]);
            eb[i] = src_ok ? a[i] + (double)pb[p][i] : kNeg;
            el[i] = (src_ok & (u0 + i < U)) ? a[i] + (double)pl[p][i] : kNeg;
          }
          double from_left = __shfl_up(el[K - 1], 1, 64);
          if (lane == 0) from_left = kNeg;
#pragma unroll
          for (int i = 0; i < K; ++i) {
            const double left = (i == 0) ? from_left : el[i - 1];
            const bool tgt_ok = (d >= lo[i]) & (d <= hi[i]);
            a[i] = tgt_ok ? log2_add(eb[i], left) : kNeg;
          }
          store_row<K>(alpha, d, U1s, u0, a);
          const int nrow = min(d - 1 + P, dlast);
          load_lp_row<K>(lpb, nrow, U1s, u0, pb[p]);
          load_lp_row<K>(lpl, nrow, U1s, u0, pl[p]);
        }
      }
    }
    // log2-likelihood = alpha(T-1,U) + lpb(T-1,U): the column-U value of the last diagonal
#pragma unroll
    for (int i = 0; i < K; ++i) {
      if (u0 + i == U) {
        const double ll2 = a[i] + (double)lpb[(size_t)dlast * U1s + U];
        w.ll[b] = ll2;
        costs[b] = (float)(-ll2 * kLn2);
      }
    }
  } else {
    double* beta = w.beta + plane;
    double bt[K];
    {
      float last[K];
      load_lp_row<K>(lpb, dlast, U1s, u0, last);
#pragma unroll
      for (int i = 0; i < K; ++i) bt[i] = (u0 + i == U) ? (double)last[i] : kNeg;
    }
    store_row<K>(beta, dlast, U1s, u0, bt);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int row = max(dlast - 1 - p, 0);
      load_lp_row<K>(lpb, row, U1s, u0, pb[p]);
      load_lp_row<K>(lpl, row, U1s, u0, pl[p]);
    }
    for (int d0 = dlast - 1; d0 >= 0; d0 -= P) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int d = d0 - p;
        if (d >= 0) {
          double from_right = __shfl_down(bt[0], 1, 64);
          if (lane == 63) from_right = kNeg;
          double nb[K];
#pragma unroll
          for (int i = 0; i < K; ++i) {
            const int u = u0 + i;
            const double right = (i == K - 1) ? from_right : bt[i + 1];
            const bool cell_ok = (d >= lo[i]) & (d <= hi[i]);
            const bool blank_ok = cell_ok & (d + 1 <= hi[i]);  // t+1 <= T-1
            const bool label_ok = cell_ok & (u < U);
            const double tb = blank_ok ? bt[i] + (double)pb[p][i] : kNeg;
            const double tl = label_ok ? right + (double)pl[p][i] : kNeg;
            nb[i] = cell_ok ? log2_add(tb, tl) : kNeg;
          }
#pragma unroll
          for (int i = 0; i < K; ++i) bt[i] = nb[i];
          store_row<K>(beta, d, U1s, u0, bt);
          const int nrow = max(d - P, 0);
          load_lp_row<K>(lpb, nrow, U1s, u0, pb[p]);
          load_lp_row<K>(lpl, nrow, U1s, u0, pl[p]);
        }
      }
    }
    if (lane == 0) w.ll[B + b] = bt[0];
  }
}

// ---------------------------------------------------------------------------
// k_grad: gradient w.r.t. the logits, written over them (or to `grad`).
//   dx[v] = s * ( 2^(y[v] + c2) - [v==blank] * pb - [v==label] * pl ),
//   y = x*log2(e), c2 = alpha+beta-ll - lse   (all base-2 logs)
// Rows are visited in the reverse of k_lse's order so that the rows k_lse read
// last (still in the memory-side cache) are re-read first.
// ---------------------------------------------------------------------------
template <int VEC, int L, int NC, int G>
__global__ __launch_bounds__(kRowBlock) void k_grad(const float* x, float* grad, const int32_t* __restrict__ labels,
                                                    const int32_t* __restrict__ in_len,
                                                    const int32_t* __restrict__ lab_len, int B, int maxT, int maxU1,
                                                    int V, int blank, const float* __restrict__ grad_scale, Ws w,
                                                    long long nit) {
  constexpr int R = 64 / L;
  const int lane = threadIdx.x & 63;
  const int j = lane % L;
  const int r = lane / L;
  const long long wave = (long long)blockIdx.x * (kRowBlock / 64) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (kRowBlock / 64);
  const int nchunk = V / VEC;
  const int cpl = (nchunk + L - 1) / L;
  const long long total_rows = (long long)B * maxT * maxU1;
  float sq = 0.f;

  for (long long itf = wave; itf < nit; itf += nwaves) {
    const long long it = nit - 1 - itf;
    RowPos pos[G];
    size_t roff[G];
    float c2[G], pb[G], pl[G], sc[G];
    int lab[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const long long row = (it * G + g) * R + r;
      pos[g] = decode_row(row, total_rows, maxT, maxU1, in_len, lab_len);
      roff[g] = (size_t)(pos[g].in_range ? row : 0) * (size_t)V;
      c2[g] = 0.f; pb[g] = 0.f; pl[g] = 0.f; sc[g] = 0.f; lab[g] = -1;
      if (pos[g].valid) {
        const int t = pos[g].t, u = pos[g].u, T = pos[g].T, U = pos[g].U;
        const size_t ci = ((size_t)pos[g].b * w.D + (t + u)) * w.U1s + u;
        const double a = w.alpha[ci];
        const double ll2 = w.ll[pos[g].b];
        const double a_ll = a - ll2;
        const float lse2 = w.lse[ci];
        const float lpb2 = w.lpb[ci];
        sc[g] = grad_scale ? grad_scale[pos[g].b] : 1.f;
        c2[g] = (float)(a_ll + w.beta[ci]) - lse2;
        if (t < T - 1) {
          pb[g] = fast_exp2((float)(a_ll + w.beta[ci + w.U1s]) + lpb2);
        } else if (u == U) {
          pb[g] = fast_exp2((float)a_ll + lpb2);
        }
        if (u < U) {
          int l = labels[(size_t)pos[g].b * (maxU1 - 1) + u];
          l = (l < 0 || l >= V) ? 0 : l;
          lab[g] = l;
          pl[g] = fast_exp2((float)(a_ll + w.beta[ci + w.U1s + 1]) + w.lpl[ci]);
        }
      }
    }
    for (int k0 = 0; k0 < cpl; k0 += NC) {
      float v[G][NC][VEC];
#pragma unroll
      for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
          const int c = j + L * (k0 + i);
          if (pos[g].valid && c < nchunk) {
            load_vec<VEC>(x + roff[g] + (size_t)c * VEC, v[g][i]);
          } else {
#pragma unroll
            for (int q = 0; q < VEC; ++q) v[g][i][q] = 0.f;
          }
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
          const int c = j + L * (k0 + i);
          if (pos[g].in_range && c < nchunk) {
            float o[VEC];
            if (pos[g].valid) {
#pragma unroll
              for (int q = 0; q < VEC; ++q) {
                const int e = c * VEC + q;
                float gv = fast_exp2(__builtin_fmaf(v[g][i][q], kLog2e, c2[g]));
                gv -= (e == blank) ? pb[g] : 0.f;
                gv -= (e == lab[g]) ? pl[g] : 0.f;
                gv *= sc[g];
                sq = __builtin_fmaf(gv, gv, sq);
                o[q] = gv;
              }
            } else {
#pragma unroll
              for (int q = 0; q < VEC; ++q) o[q] = 0.f;
            }
            store_vec<VEC>(grad + roff[g] + (size_t)c * VEC, o);
          }
        }
      }
    }
  }
  // per-block sum of squared gradient elements (deterministic: one slot per block)
  __shared__ float red[kRowBlock / 64];
  sq = group_sum<64>(sq);
  if (lane == 0) red[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kRowBlock / 64; ++i) s += red[i];
    w.partial[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(256) void k_init(Ws w) {
  if (threadIdx.x < kStatusInts) w.status[threadIdx.x] = 0;
  for (int i = threadIdx.x; i < kMaxRowBlocks; i += 256) w.partial[i] = 0.f;
}

// out2[0] = sum costs, out2[1] = sum of the per-block partials of the last k_grad.
__global__ __launch_bounds__(256) void k_reduce_scalars(const float* __restrict__ costs, int B, Ws w, int nblocks,
                                                        double* __restrict__ out2) {
  __shared__ double sh[2][256];
  double c = 0.0, s = 0.0;
  for (int i = threadIdx.x; i < B; i += 256) c += (double)costs[i];
  for (int i = threadIdx.x; i < nblocks; i += 256) s += (double)w.partial[i];
  sh[0][threadIdx.x] = c;
  sh[1][threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (threadIdx.x < o) {
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
  long long nit;
  int blocks;
};

RowCfg pick_row_cfg(int B, int maxT, int maxU1, int V, const void* p0, const void* p1) {
  RowCfg c;
  const uintptr_t a = reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1);
  c.vec = (V % 4 == 0 && a % 16 == 0) ? 4 : (V % 2 == 0 && a % 8 == 0) ? 2 : 1;
  const int nchunk = V / c.vec;
  c.L = 8;
  while (c.L < 64 && c.L < nchunk) c.L <<= 1;
  const int cpl = (nchunk + c.L - 1) / c.L;
  c.nc = cpl >= 3 ? 4 : cpl;
  c.g = 4 / c.nc;
  const long long rows = (long long)B * maxT * maxU1;
  const long long rows_per_it = (long long)(64 / c.L) * c.g;
  c.nit = (rows + rows_per_it - 1) / rows_per_it;
  const long long want = (c.nit + (kRowBlock / 64) - 1) / (kRowBlock / 64);
  c.blocks = (int)(want < kMaxRowBlocks ? want : kMaxRowBlocks);
  if (c.blocks < 1) c.blocks = 1;
  return c;
}

#define RNNT_DISPATCH_LNG(KERNEL, VEC, cfg, ...)                                                              \
  do {                                                                                                        \
    if ((cfg).L == 64 && (cfg).nc == 4) KERNEL<VEC, 64, 4, 1><<<(cfg).blocks, kRowBlock, 0, st>>>(__VA_ARGS__); \
    else if ((cfg).L == 64 && (cfg).nc == 2) KERNEL<VEC, 64, 2, 2><<<(cfg).blocks, kRowBlock, 0, st>>>(__VA_ARGS__); \
    else if ((cfg).L == 64) KERNEL<VEC, 64, 1, 4><<<(cfg).blocks, kRowBlock, 0, st>>>(__VA_ARGS__);           \
    else if ((cfg).L == 32) KERNEL<VEC, 32, 1, 4><<<(cfg).blocks, kRowBlock, 0, st>>>(__VA_ARGS__);           \
    else if ((cfg).L == 16) KERNEL<VEC, 16, 1, 4><<<(cfg).blocks, kRowBlock, 0, st>>>(__VA_ARGS__);           \
    else KERNEL<VEC, 8, 1, 4><<<(cfg).blocks, kRowBlock, 0, st>>>(__VA_ARGS__);                               \
  } while (0)

#define RNNT_DISPATCH(KERNEL, cfg, ...)                                  \
  do {                                                                   \
    if ((cfg).vec == 4) RNNT_DISPATCH_LNG(KERNEL, 4, cfg, __VA_ARGS__);  \
    else if ((cfg).vec == 2) RNNT_DISPATCH_LNG(KERNEL, 2, cfg, __VA_ARGS__); \
    else RNNT_DISPATCH_LNG(KERNEL, 1, cfg, __VA_ARGS__);                 \
  } while (0)

int check_shape(int B, int maxT, int maxU1, int V, int blank) {
  if (B < 1 || maxT < 1 || maxU1 < 1 || V < 1) return RNNT_MI355X_INVALID_VALUE;
  if (blank < 0 || blank >= V) return RNNT_MI355X_INVALID_VALUE;
  if (maxU1 > kMaxU1) return RNNT_MI355X_UNSUPPORTED_SIZE;
  if ((long long)B * maxT * maxU1 >= (1LL << 31)) return RNNT_MI355X_UNSUPPORTED_SIZE;
  return RNNT_MI355X_OK;
}

int launch_status() { return hipGetLastError() == hipSuccess ? RNNT_MI355X_OK : RNNT_MI355X_LAUNCH_FAILURE; }

int launch_forward(const float* logits, const int32_t* labels, const int32_t* in_len, const int32_t* lab_len, int B,
                   int maxT, int maxU1, int V, int blank, float* costs, void* workspace, hipStream_t st) {
  Ws w;
  ws_layout(workspace, B, maxT, maxU1, &w);
  k_init<<<1, 256, 0, st>>>(w);
  const RowCfg cfg = pick_row_cfg(B, maxT, maxU1, V, logits, logits);
  RNNT_DISPATCH(k_lse, cfg, logits, labels, in_len, lab_len, B, maxT, maxU1, V, blank, w, cfg.nit);
  const int grid = 2 * B;
  if (maxU1 <= 64) k_sweep<1><<<grid, 64, 0, st>>>(w, costs, in_len, lab_len, B, maxT, maxU1);
  else if (maxU1 <= 128) k_sweep<2><<<grid, 64, 0, st>>>(w, costs, in_len, lab_len, B, maxT, maxU1);
  else if (maxU1 <= 256) k_sweep<4><<<grid, 64, 0, st>>>(w, costs, in_len, lab_len, B, maxT, maxU1);
  else k_sweep<8><<<grid, 64, 0, st>>>(w, costs, in_len, lab_len, B, maxT, maxU1);
  return launch_status();
}

int launch_backward(const float* logits, float* grad, const int32_t* labels, const int32_t* in_len,
                    const int32_t* lab_len, int B, int maxT, int maxU1, int V, int blank, const float* grad_scale,
                    void* workspace, hipStream_t st) {
  Ws w;
  ws_layout(workspace, B, maxT, maxU1, &w);
  const RowCfg cfg = pick_row_cfg(B, maxT, maxU1, V, logits, grad);
  RNNT_DISPATCH(k_grad, cfg, logits, grad, labels, in_len, lab_len, B, maxT, maxU1, V, blank, grad_scale, w, cfg.nit);
  return launch_status();
}

}  // namespace

extern "C" {

int rnnt_mi355x_version(void) { return 100; }

const char* rnnt_mi355x_status_string(int status) {
  switch (status) {
    case RNNT_MI355X_OK: return "ok";
    case RNNT_MI355X_INVALID_VALUE: return "invalid value (null pointer, non-positive size or blank outside [0,V))";
    case RNNT_MI355X_UNSUPPORTED_SIZE: return "unsupported size (maxU1 > 512 or B*maxT*maxU1 >= 2^31)";
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
  const int rc = check_shape(B, maxT, maxU1, V, blank);
  if (rc != RNNT_MI355X_OK) return rc;
  if (!logits || !in_len || !lab_len || !costs || !workspace || (maxU1 > 1 && !labels))
    return RNNT_MI355X_INVALID_VALUE;
  return launch_forward(logits, labels, in_len, lab_len, B, maxT, maxU1, V, blank, costs, workspace,
                        static_cast<hipStream_t>(hip_stream));
}

int rnnt_mi355x_backward(const float* logits, float* grad, const int32_t* labels, const int32_t* in_len,
                         const int32_t* lab_len, int B, int maxT, int maxU1, int V, int blank, const float* grad_scale,
                         void* workspace, void* hip_stream) {
  const int rc = check_shape(B, maxT, maxU1, V, blank);
  if (rc != RNNT_MI355X_OK) return rc;
  if (!logits || !grad || !in_len || !lab_len || !workspace || (maxU1 > 1 && !labels))
    return RNNT_MI355X_INVALID_VALUE;
  return launch_backward(logits, grad, labels, in_len, lab_len, B, maxT, maxU1, V, blank, grad_scale, workspace,
                         static_cast<hipStream_t>(hip_stream));
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
  const int rc = rnnt_mi355x_forward(logits_inout, labels, in_len, lab_len, B, maxT, maxU1, V, blank, costs,
                                     workspace, hip_stream);
  if (rc != RNNT_MI355X_OK) return rc;
  return rnnt_mi355x_backward(logits_inout, logits_inout, labels, in_len, lab_len, B, maxT, maxU1, V, blank,
                              grad_scale, workspace, hip_stream);
}

int rnnt_mi355x_reduce_scalars(const float* costs, int B, void* workspace, double* out2, void* hip_stream) {
  if (!costs || !workspace || !out2 || B < 1) return RNNT_MI355X_INVALID_VALUE;
  // only the leading (shape-independent) part of the layout is touched here
  Ws w;
  ws_layout(workspace, B, 1, 1, &w);
  k_reduce_scalars<<<1, 256, 0, static_cast<hipStream_t>(hip_stream)>>>(costs, B, w, kMaxRowBlocks, out2);
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
