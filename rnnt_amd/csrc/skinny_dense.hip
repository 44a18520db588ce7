// Dense layers over the joint lattice for MI355X (gfx950): y[M,N] = act(x[M,K] W[K,N] + b) with a very tall
// M = B*T*(U+1) and small K, N <= 128 -- the layers after the first one of the reference's fully-connected
// joint (KerasModel.get_fc, reference rnnt/keras_model.py:83-104: TimeDistributed(Dense) with tanh on all but
// the last layer; the sample model has K = 100, N = 62).  A general GEMM library tiles such a product for a
// large N and runs it at a few percent of what the memory system allows; here the whole weight matrix sits in
// LDS, a wavefront streams 32 rows of x at a time through v_mfma_f32_32x32x2_f32 and the kernels are bound by
// reading x and writing y once.
//
//   k_dense_fwd   y = act(x W + b)                       (also the data gradient: dx = (dy * act') W^T)
//   k_dense_wgrad_direct   partial dW, db per block: X^T (dy * act') contracted over the block's rows, operands
//                 straight from global memory (whole rows per wavefront load); db = column sums on the vector unit
//   k_dense_wgrad_reduce   sums the partials in block order (no atomics: bit-reproducible)
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stddef.h>
#include <stdint.h>

#include "../../include/rnnt_mi355x.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxDim = 128;
constexpr int kWgradBlocks = 1024;

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int mm_row(int r, int lane) { return 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3); }

// 4 consecutive floats p[0..3] of a row, zero where col + i >= ncol or !row_ok.  VEC = widest aligned access.
template <int VEC>
__device__ __forceinline__ void load4(const float* p, int col, int ncol, bool row_ok, float (&v)[4]) {
  if (VEC == 4 && row_ok && col + 4 <= ncol) {
    const f4v t = *reinterpret_cast<const f4v*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  } else if (VEC == 2 && row_ok && col + 4 <= ncol) {
    const f2v t0 = *reinterpret_cast<const f2v*>(p), t1 = *reinterpret_cast<const f2v*>(p + 2);
    v[0] = t0[0]; v[1] = t0[1]; v[2] = t1[0]; v[3] = t1[1];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (row_ok && col + i < ncol) ? p[i] : 0.f;
  }
}

// NT consecutive floats at p (columns col..col+NT-1 of a row of ncol), zero past the end / when !row_ok
template <int NT>
__device__ __forceinline__ void loadn(const float* p, int col, int ncol, bool row_ok, bool vec_ok, float (&v)[NT]) {
  if (NT == 4 && vec_ok && row_ok && col + 4 <= ncol) {
    const f4v t = *reinterpret_cast<const f4v*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  } else if (NT == 2 && vec_ok && row_ok && col + 2 <= ncol) {
    const f2v t = *reinterpret_cast<const f2v*>(p);
    v[0] = t[0]; v[1] = t[1];
  } else {
#pragma unroll
    for (int i = 0; i < NT; ++i) v[i] = (row_ok && col + i < ncol) ? p[i] : 0.f;
  }
}

// y[m, :] = act(a[m, :] W + b), a = x or, with DACT (data gradient through a tanh layer), a = g * (1 - z^2) with
// g the upstream gradient and z the layer's forward output.  W arrives as wt[k][n] = W[k][n] (TRANS = false) or
// W[n][k] (TRANS = true: the data gradient multiplies by W^T).  Kd = contraction length, Nd = output columns.
// LDS: W zero-padded to [Kp][Np + 4].  Lane l of MFMA tile j owns output column NT*(l%32) + j.
// Resident waves per SIMD asked of the compiler: 5 with up to 64 output columns (6 spills 11 registers, and spill
// traffic shares the memory counter: K=100, N=62 276 us with 5 against 312 us with 6 once the loop's waits are
// counted), 4 with 128 columns (two 32 x 64 accumulators more).  Blocks of 8 waves share one copy of
// W in LDS: with 4-wave blocks the 55-68 KB of a 128-column W left room for two blocks = 2 waves per SIMD
// (K=100, N=100 forward 495 -> 450 us; K=64, N=40 216 -> 185 us; K=100, N=62 unchanged within noise).
#ifndef RNNT_DENSE_FWD_WPE  // 5: the counted path below needs a build without spills (spill traffic shares the counter)
#define RNNT_DENSE_FWD_WPE 5
#endif
#ifndef RNNT_DENSE_FWD_BLOCK
#define RNNT_DENSE_FWD_BLOCK 512
#endif
constexpr int kFwdBlock = RNNT_DENSE_FWD_BLOCK, kFwdWaves = kFwdBlock / 64;
template <int NT, int AVEC, bool DACT, bool TRANS>
__global__ __launch_bounds__(kFwdBlock) __attribute__((amdgpu_waves_per_eu(NT <= 2 ? (DACT ? 5 : RNNT_DENSE_FWD_WPE) : 4))) void k_dense_fwd(const float* __restrict__ a, const float* __restrict__ z,
                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                      float* __restrict__ y, long long M, int Kd, int Nd, int act,
                                                      bool yvec_ok) {
  extern __shared__ float wl[];
  constexpr int Np = 32 * NT, pitch = Np + 4;
  const int Kp = (Kd + 7) & ~7;
  for (int i = threadIdx.x; i < Kp * Np; i += kFwdBlock) {
    const int k = i / Np, n = i - k * Np;
    float v = 0.f;
    if (k < Kd && n < Nd) v = TRANS ? w[(size_t)n * Kd + k] : w[(size_t)k * Nd + n];
    wl[k * pitch + n] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, l32 = lane & 31, h = lane >> 5;
  // Counted waits (the COUNTED path below).  A wave's memory operations complete in order and are waited for by
  // count.  The next tile's first loads are issued before this tile's 16 stores, so waiting for them need not wait
  // for the stores -- if the compiler can count the stores.  It cannot count conditional ones, and then "my loads
  // have arrived" turns into "nothing is outstanding": every tile waited for its predecessor's stores to land
  // (stores knocked out: 310 -> 240 us at K=100, N=62), and every wait for a load also drained the loads issued
  // after it.  So, where the shapes allow it, nothing in the loop is conditional: lanes whose NT columns lie beyond
  // Nd work on the last NT columns that exist (the same products, the same stores to the same addresses as the lane
  // that owns them), chunks beyond Kd re-read the row's last chunk (and are multiplied by the zero rows of W),
  // rounds beyond Kp and tiles beyond the last are loaded from clamped addresses and not used.
  const bool dup_ok = NT > 1 && yvec_ok && Nd % NT == 0 && Nd >= NT;  // uniform
  const int col = dup_ok ? min(NT * l32, Nd - NT) : NT * l32;
  float bv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bv[j] = (bias && col + j < Nd) ? bias[col + j] : 0.f;
  // One round = 32 columns (one 128-byte line of every row): the four 16-byte pieces a lane takes from its row's
  // line are requested back to back, so the line is fetched once instead of once per piece.  The rounds of all the
  // wave's tiles form one software pipeline: the loads of the next round (of this tile or of the wave's next one)
  // are issued before the matrix instructions of the current round, so that a wave's memory latency overlaps its
  // own arithmetic (16 registers; keeping a whole 32 x K slab ahead cost too many resident waves).  Only where
  // the registers allow it (up to 64 output columns, no activation gradient to fold in): K=100, N=62 forward
  // 441 -> 375 us; with 128 output columns or the second operand of the data gradient it spills and loses.
  // (A variant with every load unconditional and on a ring of 8 steps whose waits the compiler counts -- the form
  // k_dense_wgrad_direct below uses -- ran at the same 310-320 us: this kernel does not wait for its loads.  With its
  // stores knocked out it takes 240 us; FETCH_SIZE shows 0.91 GB read for 0.65 GB of x: a row's 128-byte line is
  // used by four loads that lie up to a step apart.)
  // (Knock-out timings of this kernel at K=100: N=62 373 us = 207 us of arithmetic alone (137 if the matrix pipe
  // never waited) + ~45 loads + ~120 stores and what does not overlap; N=100 610 us = 416 (274) + ...: the
  // matrix pipe at the float32 rate is the bound, memory is not.)
#ifdef RNNT_DENSE_NO_PIPE
  constexpr bool PIPE = false;
#else
  constexpr bool PIPE = !DACT && NT <= 2;
#endif
  const long long tiles = (M + 31) / 32;
  const long long wave0 = (long long)blockIdx.x * kFwdWaves + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * kFwdWaves;
  float nxt[4][4], nz[4][4];
  auto issue = [&](long long tile, int k0) {
    const long long mrow = tile * 32 + l32;
    const bool row_ok = mrow < M;
    const float* ap = a + (size_t)(row_ok ? mrow : 0) * Kd + 4 * h;
    const float* zp = DACT ? z + (size_t)(row_ok ? mrow : 0) * Kd + 4 * h : nullptr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kq = k0 + 8 * q;
      if (kq < Kp) {  // uniform
#ifdef RNNT_DENSE_NO_LOAD  // timing experiment (wrong results)
        for (int i = 0; i < 4; ++i) nxt[q][i] = (float)(kq + i), nz[q][i] = 0.5f;
#else
        load4<AVEC>(ap + kq, kq + 4 * h, Kd, row_ok, nxt[q]);
        if (DACT) load4<AVEC>(zp + kq, kq + 4 * h, Kd, row_ok, nz[q]);
#endif
      }
    }
  };
#ifndef RNNT_DENSE_NO_COUNTED
  constexpr bool COUNTED = !DACT && (AVEC == 4 || AVEC == 2);
  if (COUNTED && dup_ok && Kd % AVEC == 0 && Kd >= 4 && wave0 < tiles) {  // uniform
    auto issue_u = [&](long long tile, int k0) {
      const long long mrow = min(tile * 32 + l32, M - 1);
      const float* ap = a + (size_t)mrow * Kd;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = k0 + 8 * q + 4 * h;
        if constexpr (AVEC == 4) {
          const f4v t = *reinterpret_cast<const f4v*>(ap + min(c, Kd - 4));
          nxt[q][0] = t[0]; nxt[q][1] = t[1]; nxt[q][2] = t[2]; nxt[q][3] = t[3];
        } else {  // rows of whole 8-byte pieces: a chunk of four may end inside the row
          const f2v t0 = *reinterpret_cast<const f2v*>(ap + min(c, Kd - 2));
          const f2v t1 = *reinterpret_cast<const f2v*>(ap + min(c + 2, Kd - 2));
          nxt[q][0] = t0[0]; nxt[q][1] = t0[1]; nxt[q][2] = t1[0]; nxt[q][3] = t1[1];
        }
      }
    };
    issue_u(wave0, 0);
    // one tile: products, then its stores -- unconditional for a whole tile (WHOLE), row by row for the last, partial
    // one, which is peeled out of the loop: a conditional store anywhere in the loop makes its stores uncountable
    auto do_tile = [&](long long tile, auto whole) {
      const long long m0 = tile * 32;
      const long long ntile = tile + nwaves < tiles ? tile + nwaves : tile;
      f16v acc[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      for (int k0 = 0; k0 < Kp; k0 += 32) {
        float av[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) av[q][i] = (k0 + 8 * q + 4 * h + i < Kd) ? nxt[q][i] : 0.f;
        const bool more = k0 + 32 < Kp;
        issue_u(more ? tile : ntile, more ? k0 + 32 : 0);  // the next round, of this tile or of the wave's next one
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int kq = k0 + 8 * q;
          if (kq < Kp) {  // uniform; no memory operation inside
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float* wr = wl + (kq + 4 * h + i) * pitch + col;
#pragma unroll
              for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][i], wr[j], acc[j], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long m = m0 + mm_row(r, lane);
        if (!decltype(whole)::value && m >= M) continue;
        float o[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          float v = acc[j][r] + bv[j];
          if (act == 1) {
            const float e = __builtin_amdgcn_exp2f(v * 2.885390081777927f);
            v = __builtin_fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
          }
          o[j] = v;
        }
        float* yp = y + (size_t)m * Nd + col;
        if constexpr (NT == 4) {
          const f4v t = {o[0], o[1], o[2], o[3]};
          *reinterpret_cast<f4v*>(yp) = t;
        } else if constexpr (NT == 2) {
          const f2v t = {o[0], o[1]};
          *reinterpret_cast<f2v*>(yp) = t;
        }
      }
    };
    const long long full = M / 32;  // whole tiles
    long long tile = wave0;
    // (the wave's first tile is peeled as well: entered from it, the loop finds 16 stores behind the loads it waits
    // for on every path; entered from the prologue it would find none on one, and count none)
    if (tile < full) {
      do_tile(tile, std::true_type{});
      tile += nwaves;
    }
    for (; tile < full; tile += nwaves) do_tile(tile, std::true_type{});
    if (tile < tiles) do_tile(tile, std::false_type{});
    return;
  }
#endif
  if (PIPE && wave0 < tiles) issue(wave0, 0);
  for (long long tile = wave0; tile < tiles; tile += nwaves) {
    const long long m0 = tile * 32;
    f16v acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int k0 = 0; k0 < Kp; k0 += 32) {
      if (!PIPE) issue(tile, k0);
      float av[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) av[q][i] = DACT ? nxt[q][i] * (1.f - nz[q][i] * nz[q][i]) : nxt[q][i];
      if (PIPE) {  // next round: of this tile, or the first one of the wave's next tile
        if (k0 + 32 < Kp) issue(tile, k0 + 32);
        else if (tile + nwaves < tiles) issue(tile + nwaves, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kq = k0 + 8 * q;
        if (kq < Kp) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float* wr = wl + (kq + 4 * h + i) * pitch + col;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#ifdef RNNT_DENSE_NO_MFMA  // timing experiment (wrong results)
              acc[j][i] += av[q][i] * wr[j];
#else
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][i], wr[j], acc[j], 0, 0, 0);
#endif
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long m = m0 + mm_row(r, lane);
      if (m >= M) continue;
      float o[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float v = acc[j][r] + bv[j];
        if (act == 1) {
          const float e = __builtin_amdgcn_exp2f(v * 2.885390081777927f);
          v = __builtin_fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
        }
        o[j] = v;
      }
      float* yp = y + (size_t)m * Nd + col;
#ifdef RNNT_DENSE_NO_STORE  // timing experiment (wrong results)
      {
        float keep = 0.f;
        for (int j = 0; j < NT; ++j) keep += o[j];
        if (keep == 1.2345f) yp[0] = keep;
      }
      continue;
#endif
      if (NT == 4 && yvec_ok && col + 4 <= Nd) {
        const f4v t = {o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f4v*>(yp) = t;
      } else if (NT == 2 && yvec_ok && col + 2 <= Nd) {
        const f2v t = {o[0], o[1]};
        *reinterpret_cast<f2v*>(yp) = t;
      } else {
#pragma unroll
        for (int j = 0; j < NT; ++j)
          if (col + j < Nd) yp[j] = o[j];
      }
    }
  }
}

// Partial weight gradient of one block over its chunk of rows: P[k][n] = sum_m X(m,k) G(m,n) for k < K and
// P[K][n] = sum_m G(m,n) (bias), G = dy or, with DACT, dy * (1 - y^2).  partial: [gridDim.x][32 * KT][32 * NT].
//
// The weight gradient without LDS staging.  Which row of P an accumulator row stands for, and which column an
// accumulator column, is free: with KV accumulators down and NV across, row r of accumulator i is k = KV*r + i and
// column c of accumulator j is n = NV*CS*c + NV*cs + j (cs = the wave's column share when 128 columns are split
// over CS = 2 waves).  Lane (r, h) then needs x(m, KV*r .. KV*r+KV-1) and G(m, its NV columns) of row
// m = m0 + 2*step + h: ONE 16-byte and one 8-byte load of consecutive addresses -- a wavefront reads two whole
// rows of x and of G per step, in the order they lie in memory -- feed KV*NV matrix instructions.  No LDS, no
// barrier, no transposition; a ring of D steps keeps the loads ahead of their use.  The bias gradient is the
// column sum of G: two additions per step on the vector unit instead of a row of ones in the product.
// Every wavefront (pair) works on its own range of rows; the 4 / CS partial sums of a block meet in LDS at the end.
// (Staged through LDS, k_dense_wgrad_lds, K=100 N=62 took 385 us: 137 us with neither the global loads nor the matrix
// instructions -- LDS traffic and barriers -- 306 us without the loads, 260 us without the matrix instructions.)
#ifndef RNNT_WGRAD_DEPTH
#define RNNT_WGRAD_DEPTH 8
#endif
template <int KV, int NV, int CS, bool DACT, bool VEC>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(KV * NV >= 8 ? 2 : 3))) void k_dense_wgrad_direct(const float* __restrict__ x, const float* __restrict__ yout,
                                                               const float* __restrict__ dy, float* __restrict__ partial,
                                                               long long M, int K, int N, int KT, long long rows_per_group) {
  constexpr int D = RNNT_WGRAD_DEPTH, GROUPS = 4 / CS, NACC = KV * NV * 16;
  extern __shared__ float red[];  // [CS][NACC * 64 + NV * 64]
  const int lane = threadIdx.x & 63, l32 = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
  const int cs = wv % CS, grp = wv / CS;
  const long long r0 = ((long long)blockIdx.x * GROUPS + grp) * rows_per_group;  // even
  const long long left = M - r0 < rows_per_group ? M - r0 : rows_per_group;
  const int nrows = left > 0 ? (int)left : 0, nsteps = (nrows + 1) >> 1;
  const int kcol = KV * l32, ncol = NV * CS * l32 + NV * cs;
  // columns beyond K / N: loaded from a clamped column and multiplied away
  bool kok[KV], nok[NV];
  int kc[KV], nc[NV];
#pragma unroll
  for (int i = 0; i < KV; ++i) kok[i] = kcol + i < K, kc[i] = kok[i] ? kcol + i : 0;
#pragma unroll
  for (int j = 0; j < NV; ++j) nok[j] = ncol + j < N, nc[j] = nok[j] ? ncol + j : 0;

  // this lane's rows are r0 + h, r0 + h + 2, ...: `last` is the step of the last one
  const bool has = nrows > h;
  const int last = has ? (nrows - 1 - h) >> 1 : 0;
  const size_t row0 = has ? (size_t)(r0 + h) : 0;
  const float* xb = x + row0 * K;
  const float* gb = dy + row0 * N;
  const float* yb = DACT ? yout + row0 * N : nullptr;
  float xa[D][KV], ga[D][NV], ya[D][NV];
  auto load_step = [&](int st, float (&xv)[KV], float (&gv)[NV], float (&yv)[NV]) {
    // steps beyond the lane's last row re-read that row (a row that exists); their values are dropped at use
    const unsigned stc = (unsigned)(st < last ? st : last);
    const float* xp = xb + (size_t)stc * (unsigned)(2 * K);
    const float* gp = gb + (size_t)stc * (unsigned)(2 * N);
    const float* yp = DACT ? yb + (size_t)stc * (unsigned)(2 * N) : nullptr;
    if (VEC) {  // K % KV == 0, N % NV == 0, aligned: a lane's columns are all inside or all outside
      if (KV == 4) {
        const f4v t = *reinterpret_cast<const f4v*>(xp + kc[0]);
        xv[0] = t[0]; xv[1] = t[1]; xv[2 % KV] = t[2]; xv[3 % KV] = t[3];
      } else if (KV == 2) {
        const f2v t = *reinterpret_cast<const f2v*>(xp + kc[0]);
        xv[0] = t[0]; xv[1 % KV] = t[1];
      } else {
        xv[0] = xp[kc[0]];
      }
      if (NV == 2) {
        const f2v t = *reinterpret_cast<const f2v*>(gp + nc[0]);
        gv[0] = t[0]; gv[1 % NV] = t[1];
        if (DACT) {
          const f2v u = *reinterpret_cast<const f2v*>(yp + nc[0]);
          yv[0] = u[0]; yv[1 % NV] = u[1];
        }
      } else {
        gv[0] = gp[nc[0]];
        if (DACT) yv[0] = yp[nc[0]];
      }
    } else {
#pragma unroll
      for (int i = 0; i < KV; ++i) xv[i] = xp[kc[i]];
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        gv[j] = gp[nc[j]];
        if (DACT) yv[j] = yp[nc[j]];
      }
    }
  };

  f16v acc[KV][NV];
  float bsum[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    bsum[j] = 0.f;
#pragma unroll
    for (int i = 0; i < KV; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  // The ring fills during a first pass of D steps that multiply zeros: with the first loads issued ahead of the
  // loop instead, the compiler orders them by kind, and the loop's first wait then has to drain half the ring on
  // every pass.
#pragma unroll
  for (int d = 0; d < D; ++d) {
#pragma unroll
    for (int i = 0; i < KV; ++i) xa[d][i] = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) ga[d][j] = ya[d][j] = 0.f;
  }
  for (int s = -D; s < nsteps; s += D) {  // steps beyond nsteps hold zeros
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const bool ok = has && s >= 0 && s + d <= last;
      float g[NV], xv[KV];
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const float t = DACT ? ga[d][j] * (1.f - ya[d][j] * ya[d][j]) : ga[d][j];
        g[j] = (ok && nok[j]) ? t : 0.f;
        bsum[j] += g[j];
      }
#pragma unroll
      for (int i = 0; i < KV; ++i) xv[i] = (ok && kok[i]) ? xa[d][i] : 0.f;
#pragma unroll
      for (int i = 0; i < KV; ++i)
#pragma unroll
        for (int j = 0; j < NV; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[i], g[j], acc[i][j], 0, 0, 0);
      load_step(s + d + D, xa[d], ga[d], ya[d]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // the block's row groups add up in LDS, one after the other (fixed order)
  float* mine = red + (size_t)cs * (NACC * 64 + NV * 64);
  for (int g = 0; g < GROUPS; ++g) {
    if (grp == g) {
      if (g > 0) {
#pragma unroll
        for (int i = 0; i < KV; ++i)
#pragma unroll
          for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += mine[((i * NV + j) * 16 + r) * 64 + lane];
#pragma unroll
        for (int j = 0; j < NV; ++j) bsum[j] += mine[NACC * 64 + j * 64 + lane];
      }
      if (g < GROUPS - 1) {
#pragma unroll
        for (int i = 0; i < KV; ++i)
#pragma unroll
          for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[((i * NV + j) * 16 + r) * 64 + lane] = acc[i][j][r];
#pragma unroll
        for (int j = 0; j < NV; ++j) mine[NACC * 64 + j * 64 + lane] = bsum[j];
      }
    }
    if (GROUPS > 1) __syncthreads();
  }
  if (grp != GROUPS - 1) return;
  const int Np = 32 * NV * CS;
  float* out = partial + (size_t)blockIdx.x * (32 * KT) * Np;
#pragma unroll
  for (int i = 0; i < KV; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = KV * mm_row(r, lane) + i;
      if (k < K) {
#pragma unroll
        for (int j = 0; j < NV; ++j) out[(size_t)k * Np + ncol + j] = acc[i][j][r];
      }
    }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const float t = bsum[j] + __shfl_xor(bsum[j], 32, 64);
    if (h == 0) out[(size_t)K * Np + ncol + j] = t;
  }
}

// dw[k][n] = sum over blocks of partial[blk][k][n] (k < K), db[n] = the same for row K.  16 lanes per output
// element: lane q adds the partials of blocks q, q+16, ... in order, then the 16 sums are added pairwise in a
// fixed pattern -- the result does not depend on scheduling.
__global__ __launch_bounds__(kBlock) void k_dense_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ dw,
                                                               float* __restrict__ db, int K, int N, int Kr, int Np,
                                                               int nblocks) {
  const int i = (blockIdx.x * kBlock + threadIdx.x) >> 4, q = threadIdx.x & 15;
  const bool in_range = i < (K + 1) * N;
  const int k = in_range ? i / N : 0, n = in_range ? i - k * N : 0;
  const float* p = partial + (size_t)k * Np + n;
  float s = 0.f;
  if (in_range)
    for (int b = q; b < nblocks; b += 16) s += p[(size_t)b * Kr * Np];
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  s += __shfl_xor(s, 8, 64);
  if (in_range && q == 0) {
    if (k < K) dw[(size_t)k * N + n] = s;
    else if (db) db[n] = s;
  }
}

int check_dims(long long M, int K, int N, int act) {
  if (M < 1 || K < 1 || N < 1 || (act != 0 && act != 1)) return RNNT_MI355X_INVALID_VALUE;
  if (K > kMaxDim || N > kMaxDim || M >= (1LL << 40)) return RNNT_MI355X_UNSUPPORTED_SIZE;
  return RNNT_MI355X_OK;
}
int status() { return hipGetLastError() == hipSuccess ? RNNT_MI355X_OK : RNNT_MI355X_LAUNCH_FAILURE; }
inline int nt_for(int n) { return n <= 32 ? 1 : n <= 64 ? 2 : 4; }
inline int avec_for(const void* p, int ncol) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  return (ncol % 4 == 0 && a % 16 == 0) ? 4 : (ncol % 2 == 0 && a % 8 == 0) ? 2 : 1;
}
inline bool nvec_ok(const void* p, int ncol, int nt) {
  return nt == 1 || (ncol % nt == 0 && reinterpret_cast<uintptr_t>(p) % (sizeof(float) * nt) == 0);
}
// Persistent grid of the forward kernel: as many blocks as are resident at once (the row tiles are dealt
// round-robin, so a second, partial round of blocks would leave most of the device idle at the end).
template <typename Kern>
inline unsigned fwd_grid(long long M, Kern kern, size_t lds) {
  const long long blocks = ((M + 31) / 32 + kFwdWaves - 1) / kFwdWaves;
  int per_cu = 0, dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
    cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kFwdBlock, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  const long long resident = (long long)per_cu * cus;
  return (unsigned)(blocks < resident ? blocks : resident);
}
inline int wgrad_blocks(long long M) {
  const long long b = (M + 255) / 256;
  return (int)(b < kWgradBlocks ? b : kWgradBlocks);
}

// y = act(a W (+b)) or the data gradient; Kd/Nd in the kernel's own orientation
template <bool DACT, bool TRANS>
int launch_fwd(const float* a, const float* z, const float* w, const float* bias, float* y, long long M, int Kd, int Nd,
               int act, hipStream_t st) {
  const int nt = nt_for(Nd);
  int avec = avec_for(a, Kd);
  if (DACT) {
    const int zvec = avec_for(z, Kd);
    avec = avec < zvec ? avec : zvec;
  }
  const bool yv = nvec_ok(y, Nd, nt);
  const size_t lds = sizeof(float) * (size_t)((Kd + 7) & ~7) * (32 * nt + 4);
#define DENSE_FWD(NT, AV)                                                                                        \
  do {                                                                                                           \
    if (lds > 48 * 1024 &&                                                                                       \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense_fwd<NT, AV, DACT, TRANS>),                    \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)                \
      return RNNT_MI355X_LAUNCH_FAILURE;                                                                         \
    const unsigned grid = fwd_grid(M, k_dense_fwd<NT, AV, DACT, TRANS>, lds);                                    \
    k_dense_fwd<NT, AV, DACT, TRANS><<<grid, kFwdBlock, lds, st>>>(a, z, w, bias, y, M, Kd, Nd, act, yv);           \
  } while (0)
#define DENSE_FWD_AV(NT)              \
  do {                                \
    if (avec == 4) DENSE_FWD(NT, 4);  \
    else if (avec == 2) DENSE_FWD(NT, 2); \
    else DENSE_FWD(NT, 1);            \
  } while (0)
  if (nt == 1) DENSE_FWD_AV(1);
  else if (nt == 2) DENSE_FWD_AV(2);
  else DENSE_FWD_AV(4);
#undef DENSE_FWD_AV
#undef DENSE_FWD
  return status();
}

}  // namespace

extern "C" {

int rnnt_mi355x_dense_workspace_bytes(long long M, int K, int N, size_t* out_bytes) {
  if (!out_bytes) return RNNT_MI355X_INVALID_VALUE;
  const int rc = check_dims(M, K, N, 0);
  if (rc != RNNT_MI355X_OK) return rc;
  const int KT = (K + 1 + 31) / 32;
  *out_bytes = sizeof(float) * (size_t)wgrad_blocks(M) * (32 * KT) * (32 * nt_for(N));
  return RNNT_MI355X_OK;
}

int rnnt_mi355x_dense_forward(const float* x, const float* w, const float* bias, long long M, int K, int N, int act,
                              float* y, void* hip_stream) {
  const int rc = check_dims(M, K, N, act);
  if (rc != RNNT_MI355X_OK) return rc;
  if (!x || !w || !y) return RNNT_MI355X_INVALID_VALUE;
  return launch_fwd<false, false>(x, nullptr, w, bias, y, M, K, N, act, static_cast<hipStream_t>(hip_stream));
}

int rnnt_mi355x_dense_backward(const float* x, const float* w, const float* y, const float* dy, long long M, int K,
                               int N, int act, float* dx, float* dw, float* db, void* workspace, void* hip_stream) {
  const int rc = check_dims(M, K, N, act);
  if (rc != RNNT_MI355X_OK) return rc;
  if (!x || !w || !dy || (act == 1 && !y) || (dw && !workspace) || (db && !dw)) return RNNT_MI355X_INVALID_VALUE;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  if (dx) {  // dx[M,K] = (dy * act') W^T: the forward kernel with the roles of K and N exchanged
    const int r = act == 1 ? launch_fwd<true, true>(dy, y, w, nullptr, dx, M, N, K, 0, st)
                           : launch_fwd<false, true>(dy, nullptr, w, nullptr, dx, M, N, K, 0, st);
    if (r != RNNT_MI355X_OK) return r;
  }
  if (dw) {
    const int nt = nt_for(N), KT = (K + 1 + 31) / 32;
    int nb = wgrad_blocks(M);
    float* partial = static_cast<float*>(workspace);
    {
      const int kv = K <= 32 ? 1 : K <= 64 ? 2 : 4, nv = nt >= 2 ? 2 : 1, cs = nt == 4 ? 2 : 1;
      const bool al = (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) |
                       (act == 1 ? reinterpret_cast<uintptr_t>(y) : 0)) % 16 == 0;
      const bool vec = al && K % kv == 0 && N % nv == 0;
      const size_t lds = sizeof(float) * (size_t)cs * (kv * nv * 16 * 64 + nv * 64);
      auto go = [&](auto kern) -> int {
        if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
          return RNNT_MI355X_LAUNCH_FAILURE;
        int per_cu = 0, dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
          cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kBlock, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        if ((long long)per_cu * cus < nb) nb = per_cu * cus;
        const int groups = nb * (4 / cs);
        long long rows = (M + groups - 1) / groups;
        rows = (rows + 1) & ~1LL;
        nb = (int)((M + rows * (4 / cs) - 1) / (rows * (4 / cs)));
        kern<<<nb, kBlock, lds, st>>>(x, y, dy, partial, M, K, N, KT, rows);
        return RNNT_MI355X_OK;
      };
      int r = RNNT_MI355X_OK;
#define DENSE_WGRAD_DIRECT3(KV, NV, CS)                                                                       \
  do {                                                                                                        \
    if (act == 1) r = vec ? go(k_dense_wgrad_direct<KV, NV, CS, true, true>) : go(k_dense_wgrad_direct<KV, NV, CS, true, false>);   \
    else r = vec ? go(k_dense_wgrad_direct<KV, NV, CS, false, true>) : go(k_dense_wgrad_direct<KV, NV, CS, false, false>);          \
  } while (0)
#define DENSE_WGRAD_DIRECT(KV)                       \
  do {                                               \
    if (nt == 1) DENSE_WGRAD_DIRECT3(KV, 1, 1);      \
    else if (nt == 2) DENSE_WGRAD_DIRECT3(KV, 2, 1); \
    else DENSE_WGRAD_DIRECT3(KV, 2, 2);              \
  } while (0)
      if (kv == 1) DENSE_WGRAD_DIRECT(1);
      else if (kv == 2) DENSE_WGRAD_DIRECT(2);
      else DENSE_WGRAD_DIRECT(4);
#undef DENSE_WGRAD_DIRECT
#undef DENSE_WGRAD_DIRECT3
      if (r != RNNT_MI355X_OK) return r;
    }
    const int total = (K + 1) * N * 16;
    k_dense_wgrad_reduce<<<(total + kBlock - 1) / kBlock, kBlock, 0, st>>>(partial, dw, db, K, N, 32 * KT, 32 * nt, nb);
  }
  return status();
}

}  // extern "C"
