/*
 * fp32 CPU restatement of the RNN-T loss path -- TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product library (rnnt_amd/csrc) never links or calls
 * it.
 *
 * What it restates: the work behind
 *     acts = tf.nn.log_softmax(acts)                       reference rnnt/base/util.py:66, :109
 *     rnnt_loss(acts, labels, input_length, label_length,
 *               blank_label=39)                            reference rnnt/base/util.py:70-72, :113-115
 * plus the gradient TensorFlow chains back through both ops during
 * model.fit_generator (reference rnnt/train.py:151-157) and the per-utterance
 * upstream factor of the Keras batch mean (reference rnnt/keras_model.py:215-217).
 *
 * rnnt_loss is the un-vendored, un-pinned third-party package
 * warprnnt_tensorflow (HawkAaron/warp-transducer); its body is not in the
 * reference tree.  This file restates the published algorithm (Graves 2012,
 * arXiv:1211.3711, cited by the reference at rnnt/base/common.py:49-52) in the
 * structure that dependency's CPU path is understood to have: threads over
 * utterances, one sequential lattice per utterance, fp32 natural-log
 * arithmetic.  PARITY: unpinned by the reference (it has no tests of this
 * path); pinned against oracle/rnnt_oracle.py (float64) which in turn is
 * pinned by the public known-answer vector, brute-force path enumeration and
 * finite differences.
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline float log_add(float a, float b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  float m = a > b ? a : b;
  return m + log1pf(expf(-fabsf(a - b)));
}

/* log-sum-exp of one row of V logits: the denominator of tf.nn.log_softmax
 * (reference util.py:66, :109). */
static float row_lse(const float* x, int V) {
  float m = x[0];
  for (int v = 1; v < V; ++v) m = x[v] > m ? x[v] : m;
  float s = 0.f;
  for (int v = 0; v < V; ++v) s += expf(x[v] - m);
  return m + logf(s);
}

/* One utterance: cost and gradient w.r.t. the raw logits (rnnt_loss at
 * reference util.py:70-72 / :113-115 chained through the log_softmax of
 * util.py:66 / :109).  x: [maxT, maxU1, V] view of this utterance; g may
 * alias x (the gradient of a row is written after the row is read). */
static int one_utt(const float* x, float* g, const int32_t* y, int T, int U, int maxT, int maxU1, int V, int blank,
                   float scale, float* cost_out) {
  const int U1 = U + 1;
  const size_t cells = (size_t)T * U1;
  float* ws = (float*)malloc(sizeof(float) * cells * 5);
  if (!ws) return 1;
  float *lse = ws, *lpb = ws + cells, *lpl = ws + 2 * cells, *al = ws + 3 * cells, *be = ws + 4 * cells;
#define ROW(t, u) (x + ((size_t)(t) * maxU1 + (u)) * (size_t)V)
#define C(t, u) ((size_t)(t) * U1 + (u))
  for (int t = 0; t < T; ++t)
    for (int u = 0; u < U1; ++u) {
      const float* r = ROW(t, u);
      float l = row_lse(r, V);
      lse[C(t, u)] = l;
      lpb[C(t, u)] = r[blank] - l;
      lpl[C(t, u)] = (u < U) ? r[y[u]] - l : -INFINITY;
    }
  /* alpha */
  al[0] = 0.f;
  for (int t = 0; t < T; ++t)
    for (int u = 0; u < U1; ++u) {
      if (t == 0 && u == 0) continue;
      float a = t > 0 ? al[C(t - 1, u)] + lpb[C(t - 1, u)] : -INFINITY;
      float c = u > 0 ? al[C(t, u - 1)] + lpl[C(t, u - 1)] : -INFINITY;
      al[C(t, u)] = log_add(a, c);
    }
  const float ll = al[C(T - 1, U)] + lpb[C(T - 1, U)];
  /* beta */
  be[C(T - 1, U)] = lpb[C(T - 1, U)];
  for (int t = T - 1; t >= 0; --t)
    for (int u = U; u >= 0; --u) {
      if (t == T - 1 && u == U) continue;
      float a = t < T - 1 ? be[C(t + 1, u)] + lpb[C(t, u)] : -INFINITY;
      float c = u < U ? be[C(t, u + 1)] + lpl[C(t, u)] : -INFINITY;
      be[C(t, u)] = log_add(a, c);
    }
  *cost_out = -ll;
  /* gradient, row by row (read row, then overwrite it) */
  for (int t = 0; t < maxT; ++t)
    for (int u = 0; u < maxU1; ++u) {
      float* gr = g + ((size_t)t * maxU1 + u) * (size_t)V;
      if (t >= T || u > U) {
        memset(gr, 0, sizeof(float) * V);
        continue;
      }
      const float* r = ROW(t, u);
      const float a = al[C(t, u)];
      const float c = a + be[C(t, u)] - ll - lse[C(t, u)];
      float pb = 0.f, pl = 0.f;
      if (t < T - 1)
        pb = expf(a + lpb[C(t, u)] + be[C(t + 1, u)] - ll);
      else if (u == U)
        pb = expf(a + lpb[C(t, u)] - ll);
      if (u < U) pl = expf(a + lpl[C(t, u)] + be[C(t, u + 1)] - ll);
      for (int v = 0; v < V; ++v) gr[v] = expf(r[v] + c);
      gr[blank] -= pb;
      if (u < U) gr[y[u]] -= pl;
      if (scale != 1.f)
        for (int v = 0; v < V; ++v) gr[v] *= scale;
    }
#undef ROW
#undef C
  free(ws);
  return 0;
}

/* costs[B] and, when grad != NULL, d(sum_b grad_scale[b]*cost_b)/d(logits)
 * into grad (grad may be the logits buffer itself).  Returns 0 on success,
 * 1 on allocation failure, 2 on invalid lengths/labels. */
int rnnt_oracle_cpu_loss_grad(const float* logits, const int32_t* labels, const int32_t* in_len, const int32_t* lab_len,
                              int B, int maxT, int maxU1, int V, int blank, const float* grad_scale, float* costs,
                              float* grad, int num_threads) {
  for (int b = 0; b < B; ++b) {
    if (in_len[b] < 1 || in_len[b] > maxT || lab_len[b] < 0 || lab_len[b] > maxU1 - 1) return 2;
    for (int u = 0; u < lab_len[b]; ++u) {
      int32_t l = labels[(size_t)b * (maxU1 - 1) + u];
      if (l < 0 || l >= V) return 2;
    }
  }
  if (blank < 0 || blank >= V) return 2;
  int err = 0;
  const size_t per = (size_t)maxT * maxU1 * V;
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(| : err)
  for (int b = 0; b < B; ++b) {
    /* cost only: the gradient goes to a throw-away buffer */
    float* g = grad ? grad + per * b : (float*)malloc(sizeof(float) * per);
    if (!g) {
      err |= 1;
      continue;
    }
    err |= one_utt(logits + per * b, g, labels + (size_t)b * (maxU1 - 1), in_len[b], lab_len[b], maxT, maxU1, V,
                   blank, grad_scale ? grad_scale[b] : 1.f, &costs[b]);
    if (!grad) free(g);
  }
  return err;
}

int rnnt_oracle_cpu_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
