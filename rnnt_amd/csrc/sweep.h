// Part of the single translation unit rnnt_mi355x.hip: k_sweep, the alpha / beta recursions over anti-diagonals.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

#include "workspace.h"
#include "device_util.h"

namespace {

// ---------------------------------------------------------------------------
// k_sweep: alpha (even blocks) and beta (odd blocks) anti-diagonal sweeps.
// One wavefront per (utterance, direction); lane l owns the K lattice columns
// u = l*K .. l*K+K-1, so one cross-lane shift (DPP, no LDS) per diagonal is the
// only communication and no barrier sits on the dependent chain.  Cells
// outside the lattice are handled by masking the log-probs that lead into or
// out of them to "log 0": their alpha/beta then evaluate to ~-1e30 by
// themselves.  The log-prob rows of the next P diagonals are kept in flight in
// a register ring.  With the ring deep enough a diagonal costs the longer of its dependent chain (add, shift,
// subtract, exp2, log2, add: ~50 ns) and of what the wavefront issues (~2 ns an instruction of any kind).
// Forms measured and rejected:
//  * the columns of a direction over two wavefronts, one column per lane, boundary value through LDS two diagonals
//    ahead, one s_barrier per diagonal: 2.5x slower (207 us against 83 us at T=400, U=100);
//  * alpha and beta in the linear domain as (float64 mantissa, int32 exponent) pairs -- multiply, ldexp, add, frexp
//    instead of exp2 / log2 on the chain, same results: one column per lane 77 us against 61 us, two columns 135 us
//    against 76 us at T=500 (the float64 ldexp / frexp steps cost more than the transcendentals they replace);
//  * two wavefronts for 65..128 label positions with the hand-over of k_sweep_wide (bands through LDS) and the
//    records read in place: 74 ns per diagonal plus a band of lead against 83 ns for two columns per lane -- 2 us in
//    50, not kept; with the re-ordered copy of k_skew_records in front it loses (the copy costs 12-24 us).
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
__device__ __forceinline__ void load_lp_diag(const float4* rec0, int d, int maxT, unsigned stride16,
                                             const int (&ucol)[K], const unsigned (&uoff16)[K], float (&pb)[K],
                                             float (&pl)[K]) {
#pragma unroll
  for (int i = 0; i < K; ++i) {
    // frame index incl. guards: (d - u) + 1 clamped into [0, maxT + 1]; subtract, max, min, one 24-bit multiply-add
    // (t1 <= maxT + 1 < 2^24, stride16 = 16 * maxU1 <= 2^14) giving a 32-bit byte offset from a uniform base
    const int t1 = min(max(d + 1 - ucol[i], 0), maxT + 1);
    const unsigned off = __umul24((unsigned)t1, stride16) + uoff16[i];  // uoff16 = 16 * min(u, maxU1-1) or a guard slot
    const float2 q = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(rec0) + off);
    pb[i] = q.x;
    pl[i] = q.y;
  }
}

// The same records reached by a running byte offset: the loads of a direction ask for consecutive diagonals, so a
// column's unclamped offset moves by one frame (stride16 bytes) per load and only the clamp is left -- an add, a
// v_med3_i32 and the load instead of subtract, max, min, multiply-add and the load (the loop costs what it issues).
// Signed 32-bit offsets: (maxT + maxU1 + 2) * maxU1 * 16 < 2^31 (the shape check).
template <int K>
struct RecCursor {
  int lin[K], lo[K], hi[K];
  __device__ __forceinline__ void init(int d, int maxT, unsigned stride16, const int (&ucol)[K], const unsigned (&uoff16)[K]) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const bool has = ucol[i] < (1 << 28);  // lanes beyond the last column read frame -1's first record ("log 0")
      lin[i] = has ? (d + 1 - ucol[i]) * (int)stride16 + (int)uoff16[i] : 0;
      lo[i] = has ? (int)uoff16[i] : 0;
      hi[i] = has ? (maxT + 1) * (int)stride16 + (int)uoff16[i] : 0;
    }
  }
  template <int DIR>  // +1: the next load is the next diagonal; -1: the one before
  __device__ __forceinline__ void load(const float4* rec0, int step16, float (&pb)[K], float (&pl)[K]) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
      int off;
      asm("v_med3_i32 %0, %1, %2, %3" : "=v"(off) : "v"(lin[i]), "v"(lo[i]), "v"(hi[i]));
      const float2 q = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(rec0) + (unsigned)off);
      pb[i] = q.x;
      pl[i] = q.y;
      lin[i] += DIR * step16;
    }
  }
};

template <int K>
__device__ __forceinline__ void store_row(double* base, int row, int u0, const double (&v)[K]) {
  double* p = base + ((unsigned)row * (64u * K) + (unsigned)u0);  // row stride 64 * K: a constant, so that the rows of
                                                                  // an unrolled group differ by immediate offsets
  if constexpr (K == 1) {
    p[0] = v[0];
  } else {
#pragma unroll
    for (int i = 0; i < K; i += 2) *reinterpret_cast<double2*>(p + i) = make_double2(v[i], v[i + 1]);
  }
}

// log2(2^x + 2^y) with the sum in float64 and the correction term in float32
// (the maximum as the bare instruction: fmax() first re-quiets an operand that came out of a DPP move -- one more
// instruction on a loop that costs what it issues; neither operand is ever a NaN here)
__device__ __forceinline__ double log2_add(double x, double y) {
  const float df = (float)(x - y);
  double m;
  asm("v_max_f64 %0, %1, %2" : "=v"(m) : "v"(x), "v"(y));
  return m + (double)fast_log2(1.0f + fast_exp2(-fabsf(df)));
}

template <int K>
__global__ __launch_bounds__(64) void k_sweep(Ws w, float* __restrict__ costs, const int32_t* __restrict__ in_len,
                                              const int32_t* __restrict__ lab_len, int B, int maxT, int maxU1,
                                              int b0) {
  // Diagonals of log-probs kept in flight ahead of the dependent chain.  A diagonal's loads take ~0.4 us to come
  // back (most hit the vector cache, every eighth goes further) while its arithmetic takes 40-60 ns per column:
  // with 4 in flight the loop waited for memory (109 / 127 ns per diagonal with one / two columns per lane);
  // 16 / 8 brought 60 / 92 ns.  (24 and 12: 64 / 94 ns.  Wider lanes are bound by their arithmetic: no change.)
  // A wavefront that is alone on its SIMD issues an instruction every ~2 ns whatever its kind, so the loop costs
  // what it issues: the records' running byte offsets (RecCursor: add, v_med3_i32, load per column instead of
  // subtract, max, min, multiply-add, load), rows of alpha / beta behind a running pointer (immediate offsets inside
  // an unrolled group) and the bare v_max_f64 took one / two columns per lane from 30 / 40 to 20 / 34 instructions
  // per diagonal and from 61 / 92 to 54 / 83 ns (one column is then at the pace of its dependent chain; deeper
  // rings, 12 or 16 diagonals for two columns, and a single wait per diagonal change nothing).
#ifndef RNNT_SWEEP_P2
#define RNNT_SWEEP_P2 8
#endif
#ifndef RNNT_SWEEP_P4
#define RNNT_SWEEP_P4 4
#endif
#ifndef RNNT_SWEEP_P8
#define RNNT_SWEEP_P8 4
#endif
  constexpr int P = K == 1 ? 16 : K == 2 ? RNNT_SWEEP_P2 : K == 4 ? RNNT_SWEEP_P4 : K == 8 ? RNNT_SWEEP_P8 : 4;
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
  constexpr unsigned U1s = 64u * K;  // = w.U1s (the launch picks K from the same rule as the layout)
  const size_t plane = (size_t)b * w.D * U1s;
  const float4* rec0 = w.rec + (size_t)b * (maxT + 2) * maxU1;  // frame -1 (guard), column 0
  const int u0 = lane * K;
  const int dlast = T - 1 + U;

  // columns beyond maxU1-1 have no record: they read the guard frame (frame index forced to -1 by a huge column)
  int ucol[K];
  unsigned uoff16[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int u = u0 + i;
    const bool has = u < maxU1;
    ucol[i] = has ? u : (1 << 28);
    uoff16[i] = has ? 16u * (unsigned)u : 0u;
  }
  const unsigned stride16 = 16u * (unsigned)maxU1;

  if (dir == 0) {
    double* alpha = w.alpha + plane;
    double a[K];
#pragma unroll
    for (int i = 0; i < K; ++i) a[i] = (u0 + i == 0) ? 0.0 : kNeg;
    store_row<K>(alpha, 0, u0, a);
    double* sp = alpha + (64u * K + (unsigned)u0);  // the next row of alpha (rows are written in order)
    constexpr bool kCursor = K <= 2;  // (wider lanes are bound by their arithmetic and short of registers)
    RecCursor<K> cur;
    if constexpr (kCursor) cur.init(0, maxT, stride16, ucol, uoff16);
    auto fetch = [&](int dd, float (&b_)[K], float (&l_)[K]) {  // the loads below ask for diagonals 0, 1, 2, ... in order
      if constexpr (kCursor) cur.template load<1>(rec0, (int)stride16, b_, l_);
      else load_lp_diag<K>(rec0, dd, maxT, stride16, ucol, uoff16, b_, l_);
    };
    double from_left = kNeg;
    // one diagonal: a <- alpha(d) from alpha(d-1) and the log-probs of diagonal d-1
    auto step = [&](int d, const float (&pb)[K], const float (&pl)[K]) {
      double el[K];
#pragma unroll
      for (int i = 0; i < K; ++i) el[i] = a[i] + (double)pl[i];
      from_left = lane_shift_up(el[K - 1], from_left);  // lane 0 keeps its "log 0": no constant to re-load
#pragma unroll
      for (int i = 0; i < K; ++i) a[i] = log2_add(a[i] + (double)pb[i], (i == 0) ? from_left : el[i - 1]);
      store_row<K>(sp, 0, 0, a);
      sp += 64 * K;
    };
    int d = 1;
    float pb[P][K], pl[P][K];
    // remainder first (its loads all requested before the first step), so that the pipelined loop below runs
    // whole groups of P
    const int rem = dlast % P;
#pragma unroll
    for (int p = 0; p < P - 1; ++p)
      if (p < rem) fetch(p, pb[p], pl[p]);
#pragma unroll
    for (int p = 0; p < P - 1; ++p)
      if (p < rem) step(1 + p, pb[p], pl[p]);
    d += rem;
    if (d <= dlast) {
#pragma unroll
      for (int p = 0; p < P; ++p) fetch(d - 1 + p, pb[p], pl[p]);
      for (; d <= dlast; d += P) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
          step(d + p, pb[p], pl[p]);
          fetch(d + p - 1 + P, pb[p], pl[p]);
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
    double* sp = beta + ((unsigned)dlast * (64u * K) + (unsigned)u0);  // rows dlast, dlast - 1, ... in order
    constexpr bool kCursor = K <= 2;
    RecCursor<K> cur;
    if constexpr (kCursor) cur.init(dlast, maxT, stride16, ucol, uoff16);
    auto fetch = [&](int dd, float (&b_)[K], float (&l_)[K]) {  // diagonals dlast, dlast - 1, ... in order
      if constexpr (kCursor) cur.template load<-1>(rec0, (int)stride16, b_, l_);
      else load_lp_diag<K>(rec0, dd, maxT, stride16, ucol, uoff16, b_, l_);
    };
    double from_right = kNeg;
    // one diagonal: bt <- beta(d) from beta(d+1) and the log-probs of diagonal d
    auto step = [&](int d, const float (&pb)[K], const float (&pl)[K]) {
      from_right = lane_shift_down(bt[0], from_right);  // lane 63 keeps its "log 0"
      double nb[K];
#pragma unroll
      for (int i = 0; i < K; ++i)
        nb[i] = log2_add(bt[i] + (double)pb[i], ((i == K - 1) ? from_right : bt[i + 1]) + (double)pl[i]);
#pragma unroll
      for (int i = 0; i < K; ++i) bt[i] = nb[i];
      store_row<K>(sp, 0, 0, bt);
      sp -= 64 * K;
    };
    int d = dlast;
    float pb[P][K], pl[P][K];
    const int rem = (dlast + 1) % P;
#pragma unroll
    for (int p = 0; p < P - 1; ++p)
      if (p < rem) fetch(dlast - p, pb[p], pl[p]);
#pragma unroll
    for (int p = 0; p < P - 1; ++p)
      if (p < rem) step(dlast - p, pb[p], pl[p]);
    d -= rem;
    if (d >= 0) {
#pragma unroll
      for (int p = 0; p < P; ++p) fetch(d - p, pb[p], pl[p]);
      for (; d >= 0; d -= P) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
          step(d - p, pb[p], pl[p]);
          fetch(d - p - P, pb[p], pl[p]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (lane == 0) w.ll[B + b] = bt[0];
  }
}

// ---------------------------------------------------------------------------
// Lattices with more than 128 label positions.  With 4, 8 or 16 columns per lane a diagonal of k_sweep costs
// 245 / 490 / 1000 ns: the columns of a lane are worked on one after the other.  Here a block has one wavefront
// per 64 columns, each at the one-column pace, and the two things that made that slow before are gone:
//  * k_skew_records first re-orders the log-probs by anti-diagonal (and writes "log 0" wherever a diagonal has no
//    cell), so a wavefront's 64 cells of a diagonal are 512 contiguous bytes instead of 64 cache lines -- the
//    compute unit's vector cache takes a line per cycle whichever wavefront asks;
//  * the one number per diagonal that a wavefront needs from its neighbour goes through LDS in bands of
//    kSweepBand diagonals, and the wavefronts run one band apart (rank r meets r barriers before its first band,
//    then one after each band), so a band's values were all written before the barrier in front of it: one block
//    barrier per band, nothing is polled.
// ---------------------------------------------------------------------------
// grid (U1s / 256 or 1, Dp / kSkewRows, B), blocks of min(U1s, 256) threads: a thread fills column u of kSkewRows
// diagonals (one diagonal per block is a launch of 10^5 blocks of a few instructions each: dispatch-bound)
// The barrier between two bands of k_sweep_wide: the hand-over values in LDS are all it orders (no wavefront reads
// global memory another one of the block wrote), so it waits for LDS only -- __syncthreads() would also wait for the
// wavefront's stores and for its ring of loads, and the loop would drain them once per band.
__device__ __forceinline__ void band_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int kSkewRows = 16;  // divides kSweepBand, hence Dp
__global__ __launch_bounds__(256) void k_skew_records(Ws w, int maxT, int maxU1) {
  const int U1s = w.U1s;
  const int u = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.z;
  const int d0 = blockIdx.y * kSkewRows;
  const float4* rb = w.rec + ((size_t)b * (maxT + 2) + 1) * maxU1 + u;
  float2 v[kSkewRows];
#pragma unroll
  for (int i = 0; i < kSkewRows; ++i) {  // all the loads first
    const int t = d0 + i - u;
    v[i] = make_float2(kNegF, kNegF);
    if (u < maxU1 && t >= 0 && t < maxT) v[i] = *reinterpret_cast<const float2*>(rb + (size_t)t * maxU1);
  }
  float2* out = w.lp2 + ((size_t)b * w.Dp + d0) * U1s + u;
#pragma unroll
  for (int i = 0; i < kSkewRows; ++i) out[(size_t)i * U1s] = v[i];
}

__global__ __launch_bounds__(1024) void k_sweep_wide(Ws w, float* __restrict__ costs, const int32_t* __restrict__ in_len,
                                                     const int32_t* __restrict__ lab_len, int B, int maxT, int maxU1,
                                                     int b0) {
#ifndef RNNT_WIDE_P
#define RNNT_WIDE_P 32
#endif
  constexpr int P = RNNT_WIDE_P, DB = kSweepBand;
  static_assert(DB % P == 0 && DB >= P && DB % kSkewRows == 0, "a band is whole groups of the ring");
  // Two things set a diagonal's time here, and both had to go (T=500, B=32, ns per diagonal at U+1 = 64 / 101 / 200 /
  // 400, i.e. 1 / 2 / 4 / 8 wavefronts: 101 / 107 / 117 / 176 before, 67 / 73 / 82 / 160 now; k_sweep<1> runs 61):
  //  * the re-ordered log-probs are read once each, from wherever k_skew_records left them (another XCD's L2, the
  //    memory-side cache): ~1.6 us a load, so a ring of 16 diagonals bounds a diagonal at 100 ns whatever else it
  //    does.  The ring is 32 deep (64 would overflow the 6-bit vmcnt the waits are counted in: 136 ns), and the
  //    barrier between bands waits for LDS only, in a loop that holds nothing but whole bands -- the waits stay
  //    counted and the ring is never drained;
  //  * one wavefront issues an instruction every ~2.2 ns whatever its kind (44 instructions, scalar address
  //    arithmetic included, cost 94 ns however deep the ring).  Per diagonal now 24: the hand-over value is stored by
  //    every lane through a per-lane LDS index (the boundary lane's points into the hand-over array, the others' into
  //    a junk area: no exec-mask switch); the neighbour's value is read by every lane from one address, two diagonals
  //    ahead, straight into the register the DPP shift leaves in the boundary lane (no readlane / move pairs); rows
  //    of the log-probs and of alpha / beta are reached by running pointers (in bands that cannot leave the arrays)
  //    instead of clamped 64-bit products.
  // With 8 wavefronts and more the compute unit's store path is the bound (8 x 512 B of alpha or beta per diagonal).
  constexpr int kNegOff = 16 * 2 * DB;             // [wavefront][DB] "log 0": what rank 0 reads in place of a neighbour
  constexpr int kJunkOff = kNegOff + 16 * DB;      // [wavefront][64 + DB]: where the other 63 lanes' stores land
  __shared__ double lds[kJunkOff + 16 * (64 + DB)];  // first: [boundary between rank r and r + 1][band parity][diagonal]
  const int W = (int)(blockDim.x >> 6);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int b = b0 + (blockIdx.x >> 1);
  const int dir = blockIdx.x & 1;
  const int Traw = in_len[b], Uraw = lab_len[b];
  const bool ok = (Traw >= 1) & (Traw <= maxT) & (Uraw >= 0) & (Uraw <= maxU1 - 1);
  if (!ok) {  // the whole block leaves, before any barrier
    if (threadIdx.x == 0) {
      atomicOr(&w.hdr->flags, RNNT_MI355X_FLAG_BAD_LENGTH);
      w.ll[dir * B + b] = 0.0;
      if (dir == 0) costs[b] = __builtin_nanf("");
    }
    return;
  }
  const int T = Traw, U = Uraw;
  const unsigned U1s = (unsigned)w.U1s;
  const size_t plane = (size_t)b * w.D * U1s;
  const int u = 64 * wv + lane;  // < U1s: the block has ceil(maxU1 / 64) wavefronts
  const int dlast = T - 1 + U;
  const float2* lp = w.lp2 + (size_t)b * w.Dp * U1s + u;
  const int rowmax = w.Dp - 1;
  const int rank = dir == 0 ? wv : W - 1 - wv;  // the wavefront that leads owns the columns the recursion starts in
  float pb[P], pl[P];
  auto load_lp = [&](int d, float& b_, float& l_) {
    const float2 q = lp[(size_t)min(max(d, 0), rowmax) * U1s];
    b_ = q.x;
    l_ = q.y;
  };
  if (lane < DB) lds[kNegOff + wv * DB + lane] = kNeg;  // read back by this wavefront only: LDS keeps its order
  const int hin_of_rank = rank > 0 ? (rank - 1) * 2 * DB : -1;
  const int boundary_lane = dir == 0 ? 63 : 0;
  const int hout_junk = kJunkOff + wv * (64 + DB) + lane;
  auto hin_index = [&](int j) { return hin_of_rank >= 0 ? hin_of_rank + (j & 1) * DB : kNegOff + wv * DB; };
  auto hout_index = [&](int j) { return lane == boundary_lane ? (rank * 2 + (j & 1)) * DB : hout_junk; };

  if (dir == 0) {
    double* alpha = w.alpha + plane + u;
    double a = (u == 0) ? 0.0 : kNeg;
    alpha[0] = a;
    const int nbands = (dlast + DB - 1) / DB;  // diagonals 1 .. dlast
    const int nfull = dlast / DB;               // bands wholly inside the lattice; their loads stay below row Dp
    // the ring: diagonal d needs the log-probs of diagonal d - 1
    if (nbands > 0) {
#pragma unroll
      for (int p = 0; p < P; ++p) load_lp(p, pb[p], pl[p]);
    }
    const float2* lq = lp + (size_t)P * U1s;  // the next row of log-probs to request
    double* ap = alpha + U1s;                 // the next row of alpha
    auto band = [&](int j, auto whole) {
      const int d0 = 1 + j * DB;
      const int hi = hin_index(j), ho = hout_index(j);
      double hv[3];  // the neighbour's values, requested two diagonals ahead
      hv[0] = lds[hi];
      hv[1] = lds[hi + 1];
#pragma unroll
      for (int g = 0; g < DB; g += P) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int i = g + p, d = d0 + i;
          if (i + 2 < DB) hv[(i + 2) % 3] = lds[hi + i + 2];
          if (decltype(whole)::value || d <= dlast) {  // uniform
            const double el = a + (double)pl[p];
            lds[ho + i] = el;
            const double from_left = lane_shift_up(el, hv[i % 3]);
            a = log2_add(a + (double)pb[p], from_left);
            if (decltype(whole)::value) {
              *ap = a;
              ap += U1s;
            } else {
              alpha[(size_t)d * U1s] = a;
            }
          }
          if (decltype(whole)::value) {
            const float2 q = *lq;
            lq += U1s;
            pb[p] = q.x;
            pl[p] = q.y;
          } else {
            load_lp(d - 1 + P, pb[p], pl[p]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    for (int i = 0; i < rank; ++i) band_barrier();  // the lead: rank r starts r bands behind rank 0
    // Whole bands run without a branch, every pass the same loads and stores in the same order, in a loop that holds
    // nothing else: the waits for the ring are then counted (vmcnt(N)), the loop never drains its loads.
    for (int j = 0; j < nfull; ++j) {
      band(j, std::true_type{});
      band_barrier();
    }
    if (nbands > nfull) {  // an utterance's last, partial band tests each diagonal
      band(nfull, std::false_type{});
      band_barrier();
    }
    for (int i = rank; i < W - 1; ++i) band_barrier();  // every wavefront meets nbands + W - 1 barriers
    if (u == U) {
      const double ll2 = a + (double)w.rec[((size_t)b * (maxT + 2) + T) * maxU1 + U].x;  // frame T-1 is at index T
      w.ll[b] = ll2;
      costs[b] = (float)(-ll2 * kLn2);
    }
  } else {
    double* beta = w.beta + plane + u;
    double bt = (u == U) ? 0.0 : kNeg;  // virtual diagonal dlast + 1
    const int nbands = (dlast + 1 + DB - 1) / DB;  // diagonals dlast .. 0
    const int nfull = max(dlast - P + 1, 0) / DB;   // bands whose requests (P rows ahead) stay at or above row 0
#pragma unroll
    for (int p = 0; p < P; ++p) load_lp(dlast - p, pb[p], pl[p]);
    const float2* lq = lp + (size_t)max(dlast - P, 0) * U1s;
    double* bp = beta + (size_t)dlast * U1s;
    auto band = [&](int j, auto whole) {
      const int d0 = dlast - j * DB;
      const int hi = hin_index(j), ho = hout_index(j);
      double hv[3];
      hv[0] = lds[hi];
      hv[1] = lds[hi + 1];
#pragma unroll
      for (int g = 0; g < DB; g += P) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int i = g + p, d = d0 - i;
          if (i + 2 < DB) hv[(i + 2) % 3] = lds[hi + i + 2];
          if (decltype(whole)::value || d >= 0) {  // uniform
            lds[ho + i] = bt;
            const double from_right = lane_shift_down(bt, hv[i % 3]);
            bt = log2_add(bt + (double)pb[p], from_right + (double)pl[p]);
            if (decltype(whole)::value) {
              *bp = bt;
              bp -= U1s;
            } else {
              beta[(size_t)d * U1s] = bt;
            }
          }
          if (decltype(whole)::value) {
            const float2 q = *lq;
            lq -= U1s;
            pb[p] = q.x;
            pl[p] = q.y;
          } else {
            load_lp(d - P, pb[p], pl[p]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    for (int i = 0; i < rank; ++i) band_barrier();
    for (int j = 0; j < nfull; ++j) {
      band(j, std::true_type{});
      band_barrier();
    }
    for (int j = nfull; j < nbands; ++j) {  // the bands next to row 0: clamped requests, each diagonal tested
      band(j, std::false_type{});
      band_barrier();
    }
    for (int i = rank; i < W - 1; ++i) band_barrier();
    if (u == 0) w.ll[B + b] = bt;
  }
}

}  // namespace
