/*
 * rnnt_mi355x.h -- C ABI of librnnt_mi355x.so, a from-scratch RNN-Transducer
 * loss (forward-backward + gradient) for AMD MI355X (gfx950).
 *
 * The boundary this library replaces is the one call the reference makes into
 * its un-vendored dependency:
 *
 *     from warprnnt_tensorflow import rnnt_loss
 *     rnnt_loss(acts, labels, input_length, label_length, blank_label=39)
 *         reference rnnt/base/util.py:70-72   (Util.rnnt_lambda_func_v2)
 *         reference rnnt/base/util.py:113-115 (Util.rnnt_lambda_func)
 *
 * together with the tf.nn.log_softmax the reference applies just before it
 * (util.py:66, util.py:109) and the gradient TensorFlow chains back through
 * both during training (rnnt/train.py:151-157), including the per-utterance
 * upstream factor of the Keras batch mean (rnnt/keras_model.py:215-217).
 *
 * Conventions
 *   - every pointer except `host_flags` and `out_bytes` is a DEVICE pointer;
 *   - logits: float32 [B, maxT, maxU1, V], contiguous, V fastest
 *     (acts of util.py:63 / :109 -- raw joint logits or log-probabilities:
 *     log-softmax is idempotent so both give the same costs and gradients);
 *   - labels: int32 [B, maxU1-1]; in_len, lab_len: int32 [B]
 *     (dtypes of reference rnnt/keras_model.py:125-127);
 *   - blank: any index in [0, V) (util.py:72 hard-codes 39);
 *   - costs: float32 [B], cost_b = -log P(labels_b | logits_b);
 *   - the gradient is d(sum_b grad_scale[b]*cost_b)/d(logits), i.e. it already
 *     contains the log-softmax backward; cells with t >= in_len[b] or
 *     u > lab_len[b] receive exact zeros;
 *   - the caller owns every buffer including the workspace; no entry point
 *     allocates or frees device memory, all work is enqueued on `hip_stream`
 *     (a hipStream_t, NULL = default stream; its device must be the current
 *     one) and, except rnnt_mi355x_read_status / rnnt_mi355x_read_stats, none
 *     synchronises; the only state the library keeps between calls is the
 *     default pruning threshold of rnnt_mi355x_set_prune_log2 (used only by
 *     calls that pass NaN as their own) and, per device, the address of the
 *     built-in workspace of rnnt_mi355x_loss_grad_static;
 *   - prune_log2 (last argument of every entry point that writes a gradient):
 *     the gradient pass writes an element as exact zero without computing it
 *     when it can bound its magnitude by grad_scale * 2^prune_log2 from the
 *     lattice state alone (see rnnt_mi355x_set_prune_log2).  Accepted:
 *     [-inf, -16]; -inf computes every element; NaN (the C macro NAN) = the
 *     process default.  A per-call input: concurrent calls with different
 *     thresholds do not affect each other;
 *   - every function returns an rnnt_mi355x_status; nothing throws or aborts.
 *
 * Limits: 1 <= maxU1 <= 1024, B*maxT*maxU1 < 2^31, (maxT+maxU1+32)*maxU1 < 2^27,
 * V >= 1.  Element offsets into logits are 64-bit.
 */
#ifndef RNNT_MI355X_H_
#define RNNT_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  RNNT_MI355X_OK = 0,
  RNNT_MI355X_INVALID_VALUE = 1,    /* null pointer, non-positive size, blank outside [0,V) ... */
  RNNT_MI355X_UNSUPPORTED_SIZE = 2, /* beyond a limit stated in this header */
  RNNT_MI355X_LAUNCH_FAILURE = 3,   /* HIP reported an error at launch/copy */
  RNNT_MI355X_BAD_INPUT_DATA = 4    /* device-side validation failed (see read_status) */
} rnnt_mi355x_status;

/* bits of the device-side validation word returned by rnnt_mi355x_read_status */
#define RNNT_MI355X_FLAG_BAD_LENGTH 1 /* in_len[b] outside [1,maxT] or lab_len[b] outside [0,maxU1-1] */
#define RNNT_MI355X_FLAG_BAD_LABEL 2  /* a used label outside [0,V) (clamped to 0 for the computation) */

/* Bytes of workspace the other entry points need for a batch of this shape.
 * Replaces the scratch allocation the dependency's op does internally behind
 * rnnt_loss (call sites reference util.py:70-72, 113-115). */
int rnnt_mi355x_workspace_bytes(int B, int maxT, int maxU1, size_t* out_bytes);

/* costs[b] = -log P(labels_b | logits_b): the value rnnt_loss returns at
 * reference util.py:71 / :114 (then reshaped to [B] at util.py:74 / :117).
 * Leaves the lattice state in `workspace` for a later rnnt_mi355x_backward*. */
int rnnt_mi355x_forward(const float* logits, const int32_t* labels, const int32_t* in_len, const int32_t* lab_len,
                        int B, int maxT, int maxU1, int V, int blank, float* costs, void* workspace, void* hip_stream);

/* grad = d(sum_b grad_scale[b]*cost_b)/d(logits) from the state a preceding
 * rnnt_mi355x_forward on the SAME logits left in `workspace`.  `grad` may be
 * the logits buffer itself (in place).  grad_scale: [B] floats or NULL (= 1);
 * 1/B reproduces the Keras mean of reference keras_model.py:215-217.  This is
 * the gradient registered on the dependency's op, the multiply by the
 * upstream gradient and the log_softmax backward of util.py:66 / :109 in one
 * pass. */
int rnnt_mi355x_backward(const float* logits, float* grad, const int32_t* labels, const int32_t* in_len,
                         const int32_t* lab_len, int B, int maxT, int maxU1, int V, int blank, const float* grad_scale,
                         void* workspace, void* hip_stream, float prune_log2);

/* Same, overwriting the logits with their gradient. */
int rnnt_mi355x_backward_inplace(float* logits_inout, const int32_t* labels, const int32_t* in_len,
                                 const int32_t* lab_len, int B, int maxT, int maxU1, int V, int blank,
                                 const float* grad_scale, void* workspace, void* hip_stream, float prune_log2);

/* rnnt_mi355x_forward / rnnt_mi355x_backward on tensors whose utterances are
 * `*_batch_stride` ELEMENTS apart (>= maxT*maxU1*V; inside an utterance the
 * layout stays [maxT, maxU1, V] contiguous).  This is how the crop of the
 * first two frames, `acts[:, 2:, :, :]` at reference util.py:63 (and
 * `y_trans[:, 2:, :]` at util.py:101), is taken without copying the tensor:
 * pass the address of frame 2 of utterance 0, the uncropped batch stride and
 * maxT = T0 - 2.  `grad` may alias `logits` when both strides are equal. */
int rnnt_mi355x_forward_strided(const float* logits, size_t logits_batch_stride, const int32_t* labels,
                                const int32_t* in_len, const int32_t* lab_len, int B, int maxT, int maxU1, int V,
                                int blank, float* costs, void* workspace, void* hip_stream);
int rnnt_mi355x_backward_strided(const float* logits, size_t logits_batch_stride, float* grad,
                                 size_t grad_batch_stride, const int32_t* labels, const int32_t* in_len,
                                 const int32_t* lab_len, int B, int maxT, int maxU1, int V, int blank,
                                 const float* grad_scale, void* workspace, void* hip_stream, float prune_log2);

/* The crop itself as an argument: logits_full / grad_full are dense
 * [B, T0, maxU1, V] tensors and the lattice uses frames t_offset .. T0-1 of every
 * utterance (`acts[:, shift:, :, :]`, `input_length -= shift` at reference
 * util.py:62-64; in_len counts frames of the crop, the workspace is the one of
 * rnnt_mi355x_workspace_bytes(B, T0 - t_offset, maxU1)).  The backward writes
 * zeros into the t_offset skipped frames of grad_full (which may be
 * logits_full itself).  Unlike the *_strided forms this tells the library
 * that the tensor is dense, which the small-vocabulary kernels (V <= 128: the
 * reference's own 40/62-symbol vocabularies) need. */
int rnnt_mi355x_forward_cropped(const float* logits_full, const int32_t* labels, const int32_t* in_len,
                                const int32_t* lab_len, int B, int T0, int t_offset, int maxU1, int V, int blank,
                                float* costs, void* workspace, void* hip_stream);
int rnnt_mi355x_backward_cropped(const float* logits_full, float* grad_full, const int32_t* labels,
                                 const int32_t* in_len, const int32_t* lab_len, int B, int T0, int t_offset,
                                 int maxU1, int V, int blank, const float* grad_scale, void* workspace,
                                 void* hip_stream, float prune_log2);

/* forward + backward_inplace in one call: what one training step of
 * reference rnnt/train.py:151-157 spends in the loss layer. */
int rnnt_mi355x_loss_grad(float* logits_inout, const int32_t* labels, const int32_t* in_len, const int32_t* lab_len,
                          int B, int maxT, int maxU1, int V, int blank, const float* grad_scale, float* costs,
                          void* workspace, void* hip_stream, float prune_log2);

/* rnnt_mi355x_loss_grad with optional hipEvent_t handles recorded on
 * `hip_stream` around its kernels: events4[0] before the log-sum-exp pass,
 * [1] between it and the lattice sweep, [2] between the sweep and the gradient
 * pass, [3] after the gradient pass (events4 NULL, or any entry NULL = not
 * recorded).  A measurement aid for callers that want each pass timed on the
 * launch stream; it has no counterpart in the reference. */
int rnnt_mi355x_loss_grad_events(float* logits_inout, const int32_t* labels, const int32_t* in_len,
                                 const int32_t* lab_len, int B, int maxT, int maxU1, int V, int blank,
                                 const float* grad_scale, float* costs, void* workspace, void* hip_stream,
                                 float prune_log2, void* const* events4);

/* rnnt_mi355x_loss_grad for callers that cannot allocate or own device
 * scratch memory: code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  It has no counterpart in the reference.  The
 * workspace is an array inside the library's code object
 * (RNNT_MI355X_STATIC_WORKSPACE_BYTES; every GPU the library runs on holds one
 * copy until the process ends), so the call takes no workspace and calls no
 * hipMalloc / hipFree or any other allocating routine.  Shapes whose
 * rnnt_mi355x_workspace_bytes exceeds that size are rejected with
 * RNNT_MI355X_UNSUPPORTED_SIZE.  Optional device outputs replace the calls
 * that would need the workspace pointer: scalars2 (2 doubles: what
 * rnnt_mi355x_reduce_scalars returns) and flags_out (int32: the validation
 * word of rnnt_mi355x_read_status); either may be NULL.  All calls on one GPU
 * share that GPU's array: calls that overlap on one GPU (two streams, two
 * threads) race; calls on one stream are ordered and need no clearing in
 * between (every word the kernels read is written by the same call).  The
 * first call on a device looks the array's address up
 * (hipGetSymbolAddress); make it outside graph capture. */
#define RNNT_MI355X_STATIC_WORKSPACE_BYTES (32u << 20)
int rnnt_mi355x_loss_grad_static(float* logits_inout, const int32_t* labels, const int32_t* in_len,
                                 const int32_t* lab_len, int B, int maxT, int maxU1, int V, int blank,
                                 const float* grad_scale, float* costs, double* scalars2, int32_t* flags_out,
                                 void* hip_stream, float prune_log2);

/* ---- fused additive joint ------------------------------------------------
 * The loss of the reference's additive-joint model, Util.rnnt_lambda_func
 * (reference rnnt/base/util.py:107-115): acts = log_softmax(y_trans[B,T,1,V] +
 * y_pred[B,1,U+1,V]) followed by rnnt_loss(acts, ...), WITHOUT forming the
 * [B,T,U+1,V] tensor.  f = y_trans [B,maxT,V] (utterances f_batch_stride
 * elements apart, >= maxT*V: pass the address of frame 2 and the uncropped
 * stride for the crop of util.py:101), g = y_pred [B,maxU1,V] contiguous.
 * The backward writes d(sum_b grad_scale[b]*cost_b)/df and /dg directly (what
 * TensorFlow obtains by reducing the [B,T,U+1,V] gradient over u and over t,
 * util.py:107-109).  Needs the larger workspace of
 * rnnt_mi355x_joint_workspace_bytes (which depends on V: it holds exp(f - max f)
 * and exp(g - max g), the operands of the matrix-core contractions the sums over
 * V, u and t are computed as); V <= 4096.  rnnt_mi355x_reduce_scalars
 * after a joint backward returns the cost sum and 0 for the norm. */
int rnnt_mi355x_joint_workspace_bytes(int B, int maxT, int maxU1, int V, size_t* out_bytes);
int rnnt_mi355x_joint_forward(const float* f, size_t f_batch_stride, const float* g, const int32_t* labels,
                              const int32_t* in_len, const int32_t* lab_len, int B, int maxT, int maxU1, int V,
                              int blank, float* costs, void* workspace, void* hip_stream);
int rnnt_mi355x_joint_backward(const float* f, size_t f_batch_stride, const float* g, const int32_t* labels,
                               const int32_t* in_len, const int32_t* lab_len, int B, int maxT, int maxU1, int V,
                               int blank, const float* grad_scale, float* df, size_t df_batch_stride, float* dg,
                               void* workspace, void* hip_stream, float prune_log2);

/* ---- first layer of the fully-connected joint --------------------------------
 * The reference's FC-joint model tiles the encoder output [B,T,He] and the
 * prediction-network output [B,U+1,Hd] into [B,T,U+1,He+Hd]
 * (Util.concatenate_lambda, reference rnnt/base/util.py:76-90) and applies
 * Dense(H, tanh) to it (first layer of KerasModel.get_fc, reference
 * rnnt/keras_model.py:83-104).  Because that layer is linear in the
 * concatenation, the caller computes pe = enc * W1[:He] ([B,T,H]) and
 * pd = dec * W1[He:] + b1 ([B,U1,H]) with two small matrix products and these
 * entry points do the rest without the tiled tensor:
 *   forward : h[b,t,u,:] = tanh(pe[b,t,:] + pd[b,u,:])              h: [B,T,U1,H]
 *   backward: d = dh * (1 - h^2) (h recomputed from pe, pd; dh: [B,T,U1,H]),
 *             dpe[b,t,:] = sum_u d, dpd[b,u,:] = sum_t d            (fixed summation order)
 * All tensors float32, contiguous.  The backward needs
 * rnnt_mi355x_fc_joint_workspace_bytes of scratch.  H <= 1536. */
int rnnt_mi355x_fc_joint_workspace_bytes(int B, int T, int U1, int H, size_t* out_bytes);
int rnnt_mi355x_fc_joint_forward(const float* pe, const float* pd, int B, int T, int U1, int H, float* h,
                                 void* hip_stream);
int rnnt_mi355x_fc_joint_backward(const float* pe, const float* pd, const float* dh, int B, int T, int U1, int H,
                                  float* dpe, float* dpd, void* workspace, void* hip_stream);

/* ---- Dense layers over the joint lattice -------------------------------------
 * y[M,N] = act(x[M,K] W[K,N] + bias[N]) for a very tall M = B*T*(U+1) and
 * K, N <= 128: the layers after the first one of KerasModel.get_fc (reference
 * rnnt/keras_model.py:83-104; TimeDistributed(Dense), tanh on all but the last,
 * Keras kernel layout [in, out]).  act: 0 linear, 1 tanh.  All float32,
 * contiguous.  The backward takes the forward output y (used when act == 1)
 * and the upstream gradient dy[M,N] and writes dx[M,K] (skipped when NULL),
 * dw[K,N] and db[N] (skipped when dw is NULL; db may be NULL on its own);
 * dw/db need rnnt_mi355x_dense_workspace_bytes of scratch and are summed in a
 * fixed order. */
int rnnt_mi355x_dense_workspace_bytes(long long M, int K, int N, size_t* out_bytes);
int rnnt_mi355x_dense_forward(const float* x, const float* w, const float* bias, long long M, int K, int N, int act,
                              float* y, void* hip_stream);
int rnnt_mi355x_dense_backward(const float* x, const float* w, const float* y, const float* dy, long long M, int K,
                               int N, int act, float* dx, float* dw, float* db, void* workspace, void* hip_stream);

/* ---- decoder step ---------------------------------------------------------
 * out[n, :] = log_softmax(f[f_index[n], :] + g[g_index[n], :]) over V, n < n_rows:
 * the joint + Util.softmax(out, is_log=True) the reference's beam search
 * evaluates one vector at a time (reference rnnt/rnnt_decoder.py:156-158,
 * 170-171, 200-202; rnnt/base/util.py:295-309), batched over any number of
 * (encoder frame, hypothesis) pairs.  f: [f_rows,V], g: [g_rows,V] or NULL (then
 * out = log_softmax(f rows)); the index arrays are int32 [n_rows] or NULL (= row
 * n).  An index outside its tensor sets *bad_index_flag (device int32, zeroed by
 * the caller) to 1 and row 0 is used instead.  out: [n_rows,V]. */
int rnnt_mi355x_joint_log_softmax(const float* f, int f_rows, const float* g, int g_rows, const int32_t* f_index,
                                  const int32_t* g_index, int n_rows, int V, float* out, int32_t* bad_index_flag,
                                  void* hip_stream);

/* out2[0] = sum_b costs[b], out2[1] = sum of squared gradient elements of the
 * last backward on this workspace (both float64, device memory).  These are
 * the two scalars a data-parallel caller all-reduces: the batch loss of
 * reference keras_model.py:215-217 and the gradient norm the optimizer's
 * clipnorm needs (reference keras_model.py:209-213). */
int rnnt_mi355x_reduce_scalars(const float* costs, int B, void* workspace, double* out2, void* hip_stream);

/* Pruning of the gradient pass.  With gamma = P(path through (t,u) | labels)
 * the occupancy of a lattice cell, p(v) the softmax of its row and p_max the
 * largest of them, the gradient of the row is grad_scale * (gamma*p(v) - the
 * blank's / the label's transition probability at those two v).  Every element
 * is bounded by grad_scale*gamma, and every element except the blank's and
 * the label's by grad_scale*gamma*p_max.  With T = 2^log2_threshold:
 *   gamma < T            the whole row is written as exact zeros;
 *   gamma*p_max < T      (diffuse posteriors) the row is written as exact
 *                        zeros except its blank and label entries, which are
 *                        written exactly from the lattice state alone;
 * in both cases the row's logits are not read a second time, and no element
 * written as zero exceeds grad_scale*T in magnitude.  The fused additive joint
 * applies the first rule only (it reads nothing twice: pruned cells are left
 * out of its contractions), the kernels for V <= 128 whose rows are no
 * multiple of 16 bytes likewise.  Default -28 (T = 3.7e-9); accepted range
 * [-inf, -16] (-inf: every element is computed).  This sets the process-wide
 * default that calls passing NaN as their prune_log2 use; it takes effect for
 * calls made afterwards.  A tuning knob without counterpart in the reference
 * (whose dependency computes every element). */
int rnnt_mi355x_set_prune_log2(float log2_threshold);
float rnnt_mi355x_get_prune_log2(void);

/* *host_cells = lattice cells the last backward on this workspace visited,
 * *host_pruned = how many of their rows were not read a second time (pruned or
 * sparse, see rnnt_mi355x_set_prune_log2).  Runs a
 * small kernel and waits for the stream.  A measurement aid without
 * counterpart in the reference. */
int rnnt_mi355x_read_stats(void* workspace, uint64_t* host_cells, uint64_t* host_pruned, void* hip_stream);

/* Copies the device-side validation word to *host_flags (an OR of
 * RNNT_MI355X_FLAG_*), waiting for the stream.  Returns
 * RNNT_MI355X_BAD_INPUT_DATA when it is non-zero.  Corresponds to the
 * exception the dependency's op raises on malformed inputs. */
int rnnt_mi355x_read_status(const void* workspace, int32_t* host_flags, void* hip_stream);

/* Human-readable text for a status code (static storage). */
const char* rnnt_mi355x_status_string(int status);

/* ABI version of this header: major*100 + minor. */
int rnnt_mi355x_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RNNT_MI355X_H_ */
