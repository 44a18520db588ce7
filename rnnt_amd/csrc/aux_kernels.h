// Part of the single translation unit rnnt_mi355x.hip: small kernels: zero-fill of cropped frames, workspace init, batch scalars.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "workspace.h"

namespace {

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
    w.hdr->n_count = n_partial;
    w.hdr->n_items_ready = 0;
    w.hdr->items_ctr = 0ull;
    w.hdr->rmw_next = 0u;
    w.hdr->partial_off = w.partial_off;
    w.hdr->count_off = w.count_off;
    w.hdr->n_cells = 0;
    w.hdr->n_pruned = 0;
  }
}

// Sums the per-block {cells, pruned cells} counts of the last gradient pass into the header (fixed order).
__global__ __launch_bounds__(1024) void k_reduce_counts(char* ws_base) {
  __shared__ unsigned long long sh[2][1024];
  Header* hdr = reinterpret_cast<Header*>(ws_base);
  const uint2* pc = reinterpret_cast<const uint2*>(ws_base + hdr->count_off);
  const int n = hdr->n_count;
  unsigned long long c = 0, p = 0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const uint2 v = pc[i];
    c += v.x;
    p += v.y;
  }
  sh[0][threadIdx.x] = c;
  sh[1][threadIdx.x] = p;
  __syncthreads();
  for (int o = 512; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    hdr->n_cells = sh[0][0];
    hdr->n_pruned = sh[1][0];
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
  // the partials as float4 (the array starts on a 256-byte boundary), four loads in flight per thread: the loop is
  // latency-bound, not bandwidth-bound (tens of thousands of entries at most)
  const float4* p4 = reinterpret_cast<const float4*>(partial);
  const int n4 = n >> 2;
  int i = threadIdx.x;
  for (; i + 3 * 1024 < n4; i += 4 * 1024) {
    const float4 a = p4[i], b4 = p4[i + 1024], c4 = p4[i + 2048], d4 = p4[i + 3072];
    s += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w);
    s += ((double)b4.x + (double)b4.y) + ((double)b4.z + (double)b4.w);
    s += ((double)c4.x + (double)c4.y) + ((double)c4.z + (double)c4.w);
    s += ((double)d4.x + (double)d4.y) + ((double)d4.z + (double)d4.w);
  }
  for (; i < n4; i += 1024) {
    const float4 a = p4[i];
    s += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w);
  }
  for (int k = (n4 << 2) + threadIdx.x; k < n; k += 1024) s += (double)partial[k];
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

}  // namespace
