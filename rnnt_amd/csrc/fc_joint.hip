// First layer of the reference's fully-connected joint network for MI355X (gfx950).
//
// The reference's FC-joint model tiles the encoder output [B,T,He] and the
// prediction-network output [B,U+1,Hd] into one [B,T,U+1,He+Hd] tensor
// (Util.concatenate_lambda, reference rnnt/base/util.py:76-90) and runs a stack
// of Dense layers over it (KerasModel.get_fc, reference rnnt/keras_model.py:
// 83-104; tanh on all but the last).  The first Dense layer is linear in the
// concatenation, so
//     tanh(concat(e_t, d_u) W1 + b1) = tanh(e_t W1[:He] + (d_u W1[He:] + b1))
// and the tiled tensor never needs to exist: two small matrix products give
// pe[B,T,H] and pd[B,U+1,H]; the kernels here do the broadcast, the tanh and,
// backwards, the two reductions of the upstream gradient.
//
//   k_fcj_fwd        h[b,t,u,:] = tanh(pe[b,t,:] + pd[b,u,:])          (write bound)
//   k_fcj_bwd        one pass over dh[B,T,U+1,H]: d = dh * (1 - h^2) with h recomputed,
//                    dpe[b,t,:] = sum_u d (through an LDS tile, fixed order),
//                    per-(b, frame chunk) partial sums of d over t
//   k_fcj_bwd_reduce dpd[b,u,:] = sum over the frame chunks, in order
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/rnnt_mi355x.h"

namespace {

constexpr int kBlock = 256;
constexpr int kFrames = 8;          // frames per block of the backward pass
constexpr int kTileFloats = 8192;   // LDS tile of the backward pass: label positions x H floats
constexpr float kTwoLog2e = 2.885390081777927f;

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

// tanh(x) = 1 - 2 / (2^(2x log2 e) + 1); exact limits at +-inf, relative error ~2e-7
__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(x * kTwoLog2e);
  return __builtin_fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
}

// rows of pd (label positions) per LDS tile, and tiles per utterance
inline int tile_rows(int U1, int H) {
  int r = kTileFloats / H;
  if (r < 1) r = 1;
  return r < U1 ? r : U1;
}

// h[b,t,i] = tanh(pe[b,t, i % H] + pd[b, i]) for i in [0, U1*H): both the output row of (b,t) and pd[b] are
// contiguous in i.  One block per (b, t); VEC = 4 when H % 4 == 0 and the pointers are 16-byte aligned.
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_fcj_fwd(const float* __restrict__ pe, const float* __restrict__ pd,
                                                    float* __restrict__ h, int T, int U1, int H) {
  const int bt = blockIdx.x;
  const int b = bt / T;
  const int n = U1 * H / VEC, hv = H / VEC;
  const float* per = pe + (size_t)bt * H;
  const float* pdb = pd + (size_t)b * U1 * H;
  float* out = h + (size_t)bt * U1 * H;
  int hi = threadIdx.x % hv;
  const int hstep = kBlock % hv;
  for (int i = threadIdx.x; i < n; i += kBlock) {
    if constexpr (VEC == 4) {
      const f4v a = *reinterpret_cast<const f4v*>(per + 4 * hi);
      const f4v c = *reinterpret_cast<const f4v*>(pdb + 4 * (size_t)i);
      f4v r;
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = tanh_fast(a[k] + c[k]);
      __builtin_nontemporal_store(r, reinterpret_cast<f4v*>(out + 4 * (size_t)i));
    } else {
      __builtin_nontemporal_store(tanh_fast(per[hi] + pdb[i]), out + i);
    }
    hi += hstep;
    if (hi >= hv) hi -= hv;
  }
}

// The same, with the work cut by position in memory instead of by (b, t): every block writes kFlatChunk
// consecutive vectors of h starting on a 32 KiB boundary, whatever U1*H is.  With one block per (b, t) the blocks
// start wherever their output row starts -- in the middle of a cache line unless U1*H*4 is a multiple of 128 --
// and every wavefront's store straddles lines (H = 100: 3.9 TB/s against 5.2+ at H = 64, 128).  Needs
// n = U1*H/VEC >= kFlatChunk, so that a block touches at most two (b, t) rows.
constexpr int kFlatChunk = 8 * kBlock;
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_fcj_fwd_flat(const float* __restrict__ pe, const float* __restrict__ pd,
                                                         float* __restrict__ h, long long total, int T, int U1,
                                                         int H) {
  const unsigned n = (unsigned)(U1 * H / VEC), hv = (unsigned)(H / VEC);
  const long long e0 = (long long)blockIdx.x * kFlatChunk;
  const long long bt0 = e0 / n;                       // uniform
  const unsigned i0 = (unsigned)(e0 - bt0 * n);
  const long long b0 = bt0 / T, b1 = (bt0 + 1) / T;   // utterance of the block's first and second (b, t) row
  unsigned hi = (i0 + threadIdx.x) % hv;              // n is a multiple of hv: moving on to the next row keeps it
  const unsigned hstep = kBlock % hv;
#pragma unroll
  for (int j = 0; j < kFlatChunk / kBlock; ++j) {
    const unsigned off = threadIdx.x + kBlock * j;
    const long long e = e0 + off;
    if (e < total) {
      unsigned it = i0 + off;
      const bool next = it >= n;
      it -= next ? n : 0u;
      const float* per = pe + (size_t)(bt0 + (next ? 1 : 0)) * H;
      const float* pdb = pd + (size_t)(next ? b1 : b0) * U1 * H;
      if constexpr (VEC == 4) {
        const f4v a = *reinterpret_cast<const f4v*>(per + 4 * hi);
        const f4v c = *reinterpret_cast<const f4v*>(pdb + 4 * (size_t)it);
        f4v r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = tanh_fast(a[k] + c[k]);
        __builtin_nontemporal_store(r, reinterpret_cast<f4v*>(h) + e);
      } else if constexpr (VEC == 2) {
        const f2v a = *reinterpret_cast<const f2v*>(per + 2 * hi);
        const f2v c = *reinterpret_cast<const f2v*>(pdb + 2 * (size_t)it);
        const f2v r = {tanh_fast(a[0] + c[0]), tanh_fast(a[1] + c[1])};
        __builtin_nontemporal_store(r, reinterpret_cast<f2v*>(h) + e);
      } else {
        __builtin_nontemporal_store(tanh_fast(per[hi] + pdb[it]), h + e);
      }
    }
    hi += hstep;
    if (hi >= hv) hi -= hv;
  }
}

// One block per (b, chunk of kFrames frames).  For every tile of `rows` label positions and every frame of the
// chunk: d = dh * (1 - tanh^2(pe + pd)) goes to an LDS tile [rows][H]; the threads keep the sums over the
// chunk's frames of their own tile elements in registers (-> partial[b][chunk][u][h]) and, after a barrier,
// thread h adds the tile's column h up in row order (-> dpe[b][t][h], accumulated over the tiles in LDS).
// Elements are handled in groups of VEC consecutive h (VEC = 4 when H % 4 == 0 and the pointers are aligned);
// NG = groups per thread and tile.
template <int NG, int VEC>
__global__ __launch_bounds__(kBlock) void k_fcj_bwd(const float* __restrict__ pe, const float* __restrict__ pd,
                                                    const float* __restrict__ dh, float* __restrict__ dpe,
                                                    float* __restrict__ partial, int T, int U1, int H, int rows,
                                                    int nchunk, int rg) {
  // tile[rows*H], dpe_acc[kFrames][rg][H] (the rows of a tile are summed in rg groups by different threads; the
  // groups are added up, in order, at the end), pe_s[kFrames*H]
  extern __shared__ float lds[];
  float* tile = lds;
  float* dpe_acc = lds + (size_t)rows * H;
  float* pe_s = dpe_acc + (size_t)kFrames * rg * H;
  const int b = blockIdx.x / nchunk, chunk = blockIdx.x - b * nchunk;
  const int t0 = chunk * kFrames, nt = min(kFrames, T - t0);
  for (int i = threadIdx.x; i < kFrames * rg * H; i += kBlock) dpe_acc[i] = 0.f;
  for (int i = threadIdx.x; i < kFrames * H; i += kBlock)
    pe_s[i] = (i < nt * H) ? pe[((size_t)b * T + t0) * H + i] : 0.f;
  __syncthreads();
  const int hv = H / VEC;
  const float* pdb = pd + (size_t)b * U1 * H;
  float* part = partial + ((size_t)b * nchunk + chunk) * U1 * H;
  auto ldv = [](const float* p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
      const f4v t = *reinterpret_cast<const f4v*>(p);
      v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
      v[0] = p[0];
    }
  };
  auto stv = [](float* p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) {
      const f4v t = {v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f4v*>(p) = t;
    } else {
      p[0] = v[0];
    }
  };
  for (int u0 = 0; u0 < U1; u0 += rows) {
    const int nr = min(rows, U1 - u0), n = nr * H / VEC;  // groups in this tile
    float acc[NG][VEC], pdv[NG][VEC];
    int hcol[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int i = threadIdx.x + k * kBlock;
      hcol[k] = (i * VEC) % H;
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[k][e] = pdv[k][e] = 0.f;
      if (i < n) ldv(pdb + (size_t)u0 * H + (size_t)i * VEC, pdv[k]);
    }
    for (int tt = 0; tt < nt; ++tt) {
      const float* src = dh + (((size_t)b * T + t0 + tt) * U1 + u0) * H;
      float dv[NG][VEC];
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        const int i = threadIdx.x + k * kBlock;
#pragma unroll
        for (int e = 0; e < VEC; ++e) dv[k][e] = 0.f;
        if (i < n) ldv(src + (size_t)i * VEC, dv[k]);
      }
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        const int i = threadIdx.x + k * kBlock;
        float pv[VEC], d[VEC];
        ldv(pe_s + tt * H + hcol[k], pv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float hh = tanh_fast(pv[e] + pdv[k][e]);
          d[e] = dv[k][e] * __builtin_fmaf(-hh, hh, 1.f);
          acc[k][e] += d[e];
        }
        if (i < n) stv(tile + (size_t)i * VEC, d);
      }
      __syncthreads();
      // column sums of the tile: rg threads per column group, each over its share of the rows (fixed order)
      const int rper = (nr + rg - 1) / rg;
      for (int idx = threadIdx.x; idx < rg * hv; idx += kBlock) {
        const int g = idx / hv, hc = (idx - g * hv) * VEC;
        float* accp = dpe_acc + ((size_t)tt * rg + g) * H + hc;
        float sum[VEC];
        ldv(accp, sum);
        const int r1 = min(nr, (g + 1) * rper);
        for (int r = g * rper; r < r1; ++r) {
          float tv[VEC];
          ldv(tile + r * H + hc, tv);
#pragma unroll
          for (int e = 0; e < VEC; ++e) sum[e] += tv[e];
        }
        stv(accp, sum);
      }
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int i = threadIdx.x + k * kBlock;
      if (i < n) stv(part + (size_t)u0 * H + (size_t)i * VEC, acc[k]);
    }
  }
  for (int i = threadIdx.x; i < nt * H; i += kBlock) {
    const int tt = i / H, hc = i - tt * H;
    float sum = 0.f;
    for (int g = 0; g < rg; ++g) sum += dpe_acc[((size_t)tt * rg + g) * H + hc];
    dpe[((size_t)b * T + t0) * H + i] = sum;
  }
}

// The same pass without a barrier per frame: thread j owns the 16-byte column group hc = j % (H/4) for good and the
// label positions g, g + G, g + 2G, ... (g = j / (H/4), G = 256 / (H/4) row groups per block), so that a pass of
// the block over G consecutive rows of dh is one contiguous run and the sum over u of a thread's own rows stays in
// its registers.  Per frame a thread adds its rows up (su), per label position it adds the frames up (acc, ->
// partial as in k_fcj_bwd); only the G partial sums per (frame, column) go through LDS, once, at the end of the
// block, and are added in order of g (fixed summation order).  Label positions are walked in tiles of G * NG rows.
template <int NG>
#ifndef RNNT_FCJ_WPE
#define RNNT_FCJ_WPE 3
#endif
#ifndef RNNT_FCJ_DEPTH  // frames whose loads are in flight ahead of the one being worked on, + 1
#define RNNT_FCJ_DEPTH 2
#endif
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(RNNT_FCJ_WPE))) void k_fcj_bwd_cols(const float* __restrict__ pe, const float* __restrict__ pd,
                                                         const float* __restrict__ dh, float* __restrict__ dpe,
                                                         float* __restrict__ partial, int T, int U1, int H,
                                                         int nchunk) {
  extern __shared__ float lds[];
  const int hv = H >> 2, G = kBlock / hv;
  float* pe_s = lds;                        // [kFrames][H]
  float* part_s = lds + kFrames * H;        // [kFrames][G][H]
  const int b = blockIdx.x / nchunk, chunk = blockIdx.x - b * nchunk;
  const int t0 = chunk * kFrames, nt = min(kFrames, T - t0);
  for (int i = threadIdx.x; i < kFrames * H; i += kBlock)
    pe_s[i] = (i < nt * H) ? pe[((size_t)b * T + t0) * H + i] : 0.f;
  __syncthreads();
  const int j = threadIdx.x;
  const bool on = j < G * hv;
  const int g = on ? j / hv : 0, hc = on ? j - g * hv : 0;
  const float* pdb = pd + (size_t)b * U1 * H + 4 * hc;
  const float* dhb = dh + ((size_t)b * T + t0) * U1 * H + 4 * hc;
  float* part = partial + ((size_t)b * nchunk + chunk) * U1 * H + 4 * hc;
  f4v su[kFrames];
#pragma unroll
  for (int tt = 0; tt < kFrames; ++tt) su[tt] = f4v{0.f, 0.f, 0.f, 0.f};
  for (int u0 = 0; u0 < U1; u0 += G * NG) {
    // every load is unconditional (rows past U1 and frames past nt re-read the last ones; what they produce is
    // dropped at use): no divergent branch around the loads, so the waits for them can be counted
    f4v pdv[NG], acc[NG];
    bool ok[NG];
    unsigned roff[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int u = u0 + k * G + g;
      ok[k] = on && u < U1;
      roff[k] = (unsigned)min(u, U1 - 1) * (unsigned)H;
      acc[k] = f4v{0.f, 0.f, 0.f, 0.f};
      pdv[k] = *reinterpret_cast<const f4v*>(pdb + roff[k]);
    }
    // the loads of frame tt + 1 are in flight while frame tt is worked on (two register sets; the scheduling
    // barriers keep the compiler from hoisting more frames than that: all eight would cost a resident block)
    constexpr int FD = RNNT_FCJ_DEPTH;
    f4v dv[FD][NG];
    auto fetch = [&](int tt, f4v (&d)[NG]) {
      const float* src = dhb + (size_t)min(tt, nt - 1) * U1 * H;
#pragma unroll
      for (int k = 0; k < NG; ++k) d[k] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(src + roff[k]));
    };
#pragma unroll
    for (int tt = 0; tt < FD - 1; ++tt) {
      fetch(tt, dv[tt]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int tt = 0; tt < kFrames; ++tt) {
      if (tt + FD - 1 < kFrames) fetch(tt + FD - 1, dv[(tt + FD - 1) % FD]);
      __builtin_amdgcn_sched_barrier(0);
      if (tt < nt) {  // uniform
        const f4v pev = *reinterpret_cast<const f4v*>(pe_s + tt * H + 4 * hc);
#pragma unroll
        for (int k = 0; k < NG; ++k) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float hh = tanh_fast(pev[e] + pdv[k][e]);
            const float d = ok[k] ? dv[tt % FD][k][e] * __builtin_fmaf(-hh, hh, 1.f) : 0.f;
            acc[k][e] += d;
            su[tt][e] += d;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < NG; ++k)
      if (ok[k]) *reinterpret_cast<f4v*>(part + (size_t)(u0 + k * G + g) * H) = acc[k];
  }
  if (on) {
#pragma unroll
    for (int tt = 0; tt < kFrames; ++tt) *reinterpret_cast<f4v*>(part_s + ((size_t)tt * G + g) * H + 4 * hc) = su[tt];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nt * H; i += kBlock) {
    const int tt = i / H, hcol = i - tt * H;
    float sum = 0.f;
    for (int gg = 0; gg < G; ++gg) sum += part_s[((size_t)tt * G + gg) * H + hcol];
    dpe[((size_t)b * T + t0) * H + i] = sum;
  }
}

// dpd[b][i] = sum over the frame chunks of partial[b][chunk][i], i in [0, U1*H)
__global__ __launch_bounds__(kBlock) void k_fcj_bwd_reduce(const float* __restrict__ partial,
                                                           float* __restrict__ dpd, int per_b, int nchunk,
                                                           long long total) {
  const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= total) return;
  const long long b = i / per_b;
  const int r = (int)(i - b * per_b);
  const float* p = partial + (size_t)b * nchunk * per_b + r;
  float s = 0.f;
  for (int c = 0; c < nchunk; ++c) s += p[(size_t)c * per_b];
  dpd[i] = s;
}

#ifndef RNNT_FCJ_NG_SMALL
#define RNNT_FCJ_NG_SMALL 4
#endif
#ifndef RNNT_FCJ_NG_LARGE
#define RNNT_FCJ_NG_LARGE 4
#endif
constexpr int kMaxH = 1536;  // tile + [kFrames][rg][H] + [kFrames][H] must fit the 160 KiB of LDS (rg = 1 when H > 512)

int check_dims(int B, int T, int U1, int H) {
  if (B < 1 || T < 1 || U1 < 1 || H < 1) return RNNT_MI355X_INVALID_VALUE;
  if (H > kMaxH) return RNNT_MI355X_UNSUPPORTED_SIZE;
  if ((long long)B * T >= (1LL << 31) || (long long)U1 * H >= (1LL << 31)) return RNNT_MI355X_UNSUPPORTED_SIZE;
  return RNNT_MI355X_OK;
}

int status() { return hipGetLastError() == hipSuccess ? RNNT_MI355X_OK : RNNT_MI355X_LAUNCH_FAILURE; }

}  // namespace

extern "C" {

int rnnt_mi355x_fc_joint_workspace_bytes(int B, int T, int U1, int H, size_t* out_bytes) {
  if (!out_bytes) return RNNT_MI355X_INVALID_VALUE;
  const int rc = check_dims(B, T, U1, H);
  if (rc != RNNT_MI355X_OK) return rc;
  const int nchunk = (T + kFrames - 1) / kFrames;
  *out_bytes = sizeof(float) * (size_t)B * nchunk * U1 * H;
  return RNNT_MI355X_OK;
}

int rnnt_mi355x_fc_joint_forward(const float* pe, const float* pd, int B, int T, int U1, int H, float* h,
                                 void* hip_stream) {
  const int rc = check_dims(B, T, U1, H);
  if (rc != RNNT_MI355X_OK) return rc;
  if (!pe || !pd || !h) return RNNT_MI355X_INVALID_VALUE;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const bool vec = H % 4 == 0 && (reinterpret_cast<uintptr_t>(pe) | reinterpret_cast<uintptr_t>(pd) |
                                  reinterpret_cast<uintptr_t>(h)) % 16 == 0;
  const bool vec2 = !vec && H % 2 == 0 && (reinterpret_cast<uintptr_t>(pe) | reinterpret_cast<uintptr_t>(pd) |
                                           reinterpret_cast<uintptr_t>(h)) % 8 == 0;
  const int vw = vec ? 4 : vec2 ? 2 : 1;
  const long long n = (long long)U1 * H / vw, total = (long long)B * T * n;
  if (n >= kFlatChunk) {  // blocks aligned in memory
    const unsigned grid = (unsigned)((total + kFlatChunk - 1) / kFlatChunk);
    if (vw == 4) k_fcj_fwd_flat<4><<<grid, kBlock, 0, st>>>(pe, pd, h, total, T, U1, H);
    else if (vw == 2) k_fcj_fwd_flat<2><<<grid, kBlock, 0, st>>>(pe, pd, h, total, T, U1, H);
    else k_fcj_fwd_flat<1><<<grid, kBlock, 0, st>>>(pe, pd, h, total, T, U1, H);
  } else if (vec) {
    k_fcj_fwd<4><<<B * T, kBlock, 0, st>>>(pe, pd, h, T, U1, H);
  } else {
    k_fcj_fwd<1><<<B * T, kBlock, 0, st>>>(pe, pd, h, T, U1, H);
  }
  return status();
}

int rnnt_mi355x_fc_joint_backward(const float* pe, const float* pd, const float* dh, int B, int T, int U1, int H,
                                  float* dpe, float* dpd, void* workspace, void* hip_stream) {
  const int rc = check_dims(B, T, U1, H);
  if (rc != RNNT_MI355X_OK) return rc;
  if (!pe || !pd || !dh || !dpe || !dpd || !workspace) return RNNT_MI355X_INVALID_VALUE;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const int nchunk = (T + kFrames - 1) / kFrames;
  const int rows = tile_rows(U1, H);
  const bool vec = H % 4 == 0 && (reinterpret_cast<uintptr_t>(pd) | reinterpret_cast<uintptr_t>(dh) |
                                  reinterpret_cast<uintptr_t>(workspace)) % 16 == 0;
  // threads per column group of the tile's column sums: as many as the block has for them, at most 4
  int rg = kBlock / (H / (vec ? 4 : 1) > 0 ? H / (vec ? 4 : 1) : 1);
  rg = rg < 1 ? 1 : rg > 4 ? 4 : rg;  // 8 measured slower at H = 100, 128 (LDS for the partial sums costs a resident block)
  const size_t lds = sizeof(float) * ((size_t)rows * H + (size_t)kFrames * rg * H + (size_t)kFrames * H);
  float* partial = static_cast<float*>(workspace);
  const int ng = (rows * H / (vec ? 4 : 1) + kBlock - 1) / kBlock;  // groups per thread and tile: <= 8 (vec) or 32
  const dim3 grid((unsigned)(B * nchunk));
#ifndef RNNT_FCJ_NO_COLS
  if (vec && reinterpret_cast<uintptr_t>(pe) % 16 == 0 && H / 4 <= kBlock) {
    const int G = kBlock / (H / 4);
    const size_t lds2 = sizeof(float) * ((size_t)kFrames * H + (size_t)kFrames * G * H);
#define FCJ_COLS(N)                                                                                           \
  do {                                                                                                        \
    if (lds2 > 48 * 1024 &&                                                                                   \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fcj_bwd_cols<N>),                                \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)            \
      return RNNT_MI355X_LAUNCH_FAILURE;                                                                      \
    k_fcj_bwd_cols<N><<<grid, kBlock, lds2, st>>>(pe, pd, dh, dpe, partial, T, U1, H, nchunk);                \
  } while (0)
    // rows per thread and tile: few tiles (pd is re-read per tile), few registers
    const int rows_per_pass = G;
    const int need = (U1 + rows_per_pass - 1) / rows_per_pass;  // rows a thread owns in all
    if (need <= RNNT_FCJ_NG_SMALL) FCJ_COLS(RNNT_FCJ_NG_SMALL);
    else FCJ_COLS(RNNT_FCJ_NG_LARGE);
#undef FCJ_COLS
    const long long total2 = (long long)B * U1 * H;
    k_fcj_bwd_reduce<<<(unsigned)((total2 + kBlock - 1) / kBlock), kBlock, 0, st>>>(partial, dpd, U1 * H, nchunk, total2);
    return status();
  }
#endif
#define FCJ_BWD(N, V)                                                                                       \
  do {                                                                                                      \
    if (lds > 48 * 1024 &&                                                                                  \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fcj_bwd<N, V>),                                \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)           \
      return RNNT_MI355X_LAUNCH_FAILURE;                                                                    \
    k_fcj_bwd<N, V><<<grid, kBlock, lds, st>>>(pe, pd, dh, dpe, partial, T, U1, H, rows, nchunk, rg);       \
  } while (0)
  if (vec) {
    if (ng <= 2) FCJ_BWD(2, 4);
    else if (ng <= 4) FCJ_BWD(4, 4);
    else FCJ_BWD(8, 4);
  } else {
    if (ng <= 4) FCJ_BWD(4, 1);
    else if (ng <= 8) FCJ_BWD(8, 1);
    else if (ng <= 16) FCJ_BWD(16, 1);
    else FCJ_BWD(32, 1);
  }
#undef FCJ_BWD
  const long long total = (long long)B * U1 * H;
  k_fcj_bwd_reduce<<<(unsigned)((total + kBlock - 1) / kBlock), kBlock, 0, st>>>(partial, dpd, U1 * H, nchunk, total);
  return status();
}

}  // extern "C"
