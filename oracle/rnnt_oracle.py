"""CPU oracle for the RNN-T (transducer) loss path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (rnnt_amd/) never does.

What it restates
----------------
The reference computes its training loss in two Keras lambdas,
``Util.rnnt_lambda_func`` (reference rnnt/base/util.py:93-117) and
``Util.rnnt_lambda_func_v2`` (reference rnnt/base/util.py:53-74):

  * crop the first ``shift = 2`` encoder frames and subtract 2 from the input
    lengths                                   (util.py:62-64, util.py:100-102)
  * form the joint ``y_trans[B,T,1,V] + y_pred[B,1,U+1,V]`` (util.py:107-109)
  * ``tf.nn.log_softmax`` over the last axis  (util.py:66, util.py:109)
  * reshape both length tensors to ``[B]``    (util.py:67-68, util.py:110-111)
  * ``rnnt_loss(acts, labels, input_length, label_length, blank_label=39)``
                                              (util.py:70-72, util.py:113-115)
  * reshape the result to ``[B]``             (util.py:74, util.py:117)

and Keras takes the batch mean of that vector through an identity loss
(reference rnnt/keras_model.py:215-217).

``rnnt_loss`` itself is NOT in the reference tree: it is the third-party
package ``warprnnt_tensorflow`` (HawkAaron/warp-transducer TensorFlow binding),
imported at util.py:70 and util.py:113, with no version pinned anywhere in the
reference (no requirements/setup/lock file).  Its published algorithm is the
transducer forward-backward of Graves, "Sequence Transduction with Recurrent
Neural Networks" (arXiv:1211.3711), the paper the reference itself cites at
rnnt/base/common.py:49-52.  This module restates that algorithm in float64
numpy.

PARITY PINNING: the reference holds no test, fixture or golden value for this
path ("parity unpinned" with respect to the reference's own tests).  The oracle
is pinned instead by (1) the two public warp-transducer known-answer vectors
(B=1,T=2,U=2,V=5: costs 4.495666774 / 5.095666774 and their gradients; and the
batch case B=2,T=4,U=2,V=3: costs 4.28065286 / 3.93843698 and its gradient), (2)
brute-force enumeration of every alignment path on tiny lattices and (3)
central finite differences of the cost; see tests/test_oracle.py.

Conventions: ``logits`` is ``[B, T, U1, V]`` (U1 = max label length + 1),
``labels`` ``[B, U1-1]``, ``input_lengths``/``label_lengths`` ``[B]``.  The
gradient returned is d(cost_b)/d(logits) *through* the log-softmax, multiplied
by ``grad_scale[b]`` (1/B reproduces the Keras batch mean, keras_model.py:216).
Because log_softmax is idempotent the same numbers come out whether raw joint
logits or already-log-softmaxed activations are passed in (util.py:66/109 pass
the latter).
"""
from __future__ import annotations

import itertools

import numpy as np

NEG_INF = -np.inf


def log_softmax(x: np.ndarray) -> np.ndarray:
    """Row log-softmax over the last axis, the float64 analogue of
    ``tf.nn.log_softmax`` at reference util.py:66 and util.py:109.  (The
    reference's only in-tree log-softmax, ``Util.softmax(is_log=True)`` at
    util.py:295-309, uses the same max-subtraction.)"""
    x = np.asarray(x, dtype=np.float64)
    m = np.max(x, axis=-1, keepdims=True)
    return x - (m + np.log(np.sum(np.exp(x - m), axis=-1, keepdims=True)))


def _lattice_logprobs(lp: np.ndarray, labels: np.ndarray, T: int, U: int, blank: int):
    """Blank and label log-probabilities of one utterance's lattice.
    lp: [T, U+1, V] log-probs.  Returns (lpb[T,U+1], lpl[T,U])."""
    lpb = lp[:T, : U + 1, blank]
    if U > 0:
        lpl = lp[:T, :U, :][np.arange(T)[:, None], np.arange(U)[None, :], labels[:U][None, :]]
    else:
        lpl = np.zeros((T, 0))
    return lpb, lpl


def alpha_beta(lpb: np.ndarray, lpl: np.ndarray):
    """Forward/backward lattice variables (Graves 2012, eqs. 16-18 and 19-21),
    i.e. what ``rnnt_loss`` (call sites util.py:70-72, 113-115) evaluates.

    alpha[0,0] = 0
    alpha[t,u] = logaddexp(alpha[t-1,u] + lpb[t-1,u], alpha[t,u-1] + lpl[t,u-1])
    beta[T-1,U] = lpb[T-1,U]
    beta[t,u]  = logaddexp(beta[t+1,u] + lpb[t,u],   beta[t,u+1] + lpl[t,u])
    ll = alpha[T-1,U] + lpb[T-1,U] = beta[0,0]
    """
    T, U1 = lpb.shape
    U = U1 - 1
    alpha = np.full((T, U1), NEG_INF)
    beta = np.full((T, U1), NEG_INF)
    alpha[0, 0] = 0.0
    for t in range(T):
        for u in range(U1):
            if t == 0 and u == 0:
                continue
            a = alpha[t - 1, u] + lpb[t - 1, u] if t > 0 else NEG_INF
            c = alpha[t, u - 1] + lpl[t, u - 1] if u > 0 else NEG_INF
            alpha[t, u] = np.logaddexp(a, c)
    beta[T - 1, U] = lpb[T - 1, U]
    for t in range(T - 1, -1, -1):
        for u in range(U, -1, -1):
            if t == T - 1 and u == U:
                continue
            a = beta[t + 1, u] + lpb[t, u] if t < T - 1 else NEG_INF
            c = beta[t, u + 1] + lpl[t, u] if u < U else NEG_INF
            beta[t, u] = np.logaddexp(a, c)
    ll = alpha[T - 1, U] + lpb[T - 1, U]
    return alpha, beta, ll


def alpha_beta_diag(lpb: np.ndarray, lpl: np.ndarray):
    """Same recursion as :func:`alpha_beta`, vectorised along anti-diagonals
    (cells with equal t+u are independent) so that BASELINE-sized lattices
    (T=500, U=100) finish in well under a second."""
    T, U1 = lpb.shape
    U = U1 - 1
    alpha = np.full((T, U1), NEG_INF)
    beta = np.full((T, U1), NEG_INF)
    alpha[0, 0] = 0.0
    with np.errstate(invalid="ignore"):
        for d in range(1, T + U):
            u = np.arange(max(0, d - (T - 1)), min(U, d) + 1)
            t = d - u
            a = np.full(u.shape, NEG_INF)
            c = np.full(u.shape, NEG_INF)
            mt = t > 0
            a[mt] = alpha[t[mt] - 1, u[mt]] + lpb[t[mt] - 1, u[mt]]
            mu = u > 0
            c[mu] = alpha[t[mu], u[mu] - 1] + lpl[t[mu], u[mu] - 1]
            alpha[t, u] = np.logaddexp(a, c)
        beta[T - 1, U] = lpb[T - 1, U]
        for d in range(T + U - 2, -1, -1):
            u = np.arange(max(0, d - (T - 1)), min(U, d) + 1)
            t = d - u
            a = np.full(u.shape, NEG_INF)
            c = np.full(u.shape, NEG_INF)
            mt = t < T - 1
            a[mt] = beta[t[mt] + 1, u[mt]] + lpb[t[mt], u[mt]]
            mu = u < U
            c[mu] = beta[t[mu], u[mu] + 1] + lpl[t[mu], u[mu]]
            beta[t, u] = np.logaddexp(a, c)
    ll = alpha[T - 1, U] + lpb[T - 1, U]
    return alpha, beta, ll


def rnnt_loss_and_grad(logits, labels, input_lengths, label_lengths, blank, grad_scale=None):
    """costs[B] and d(sum_b grad_scale[b]*cost_b)/d(logits) in float64.

    Mirrors the call ``rnnt_loss(acts, labels, input_length, label_length,
    blank_label)`` (reference util.py:70-72, 113-115) including the gradient
    TensorFlow chains through ``tf.nn.log_softmax`` (util.py:66, 109):

      dx[t,u,v] = s * ( gamma[t,u]*softmax[t,u,v]
                        - [v==blank]    * exp(alpha+lpb+beta[t+1,u]-ll)
                        - [v==y_{u+1}] * exp(alpha+lpl+beta[t,u+1]-ll) )

    with gamma = exp(alpha+beta-ll); at the terminal cell the "next beta" of
    the blank term is log 1.  Cells with t >= T_b or u > U_b get zero gradient.
    If a label equals the blank index both subtractions land on the same v
    (their sum) -- the dependency leaves that case undefined.
    """
    x = np.asarray(logits, dtype=np.float64)
    B, maxT, maxU1, V = x.shape
    labels = np.asarray(labels).reshape(B, -1)
    Tl = np.asarray(input_lengths).reshape(B).astype(np.int64)
    Ul = np.asarray(label_lengths).reshape(B).astype(np.int64)
    if grad_scale is None:
        grad_scale = np.ones(B)
    grad_scale = np.asarray(grad_scale, dtype=np.float64).reshape(B)
    costs = np.zeros(B)
    grad = np.zeros_like(x)
    for b in range(B):
        T, U = int(Tl[b]), int(Ul[b])
        assert 1 <= T <= maxT and 0 <= U <= maxU1 - 1
        lp = log_softmax(x[b, :T, : U + 1, :])
        lpb, lpl = _lattice_logprobs(lp, labels[b], T, U, blank)
        alpha, beta, ll = alpha_beta_diag(lpb, lpl)
        costs[b] = -ll
        gamma = np.exp(alpha + beta - ll)
        g = gamma[:, :, None] * np.exp(lp)
        # blank transitions
        pb = np.zeros((T, U + 1))
        if T > 1:
            pb[: T - 1, :] = np.exp(alpha[: T - 1, :] + lpb[: T - 1, :] + beta[1:, :] - ll)
        pb[T - 1, U] = np.exp(alpha[T - 1, U] + lpb[T - 1, U] - ll)
        g[:, :, blank] -= pb
        if U > 0:
            pl = np.exp(alpha[:, :U] + lpl + beta[:, 1:] - ll)
            tt, uu = np.meshgrid(np.arange(T), np.arange(U), indexing="ij")
            np.subtract.at(g, (tt, uu, labels[b][:U][None, :].repeat(T, 0)), pl)
        grad[b, :T, : U + 1, :] = grad_scale[b] * g
    return costs, grad


def rnnt_lambda_func(y_trans, y_pred, labels, input_length, label_length, blank_label=39, grad_scale=None):
    """Float64 restatement of ``Util.rnnt_lambda_func`` (reference
    util.py:93-117): crop 2 frames, broadcast-add joint, log-softmax, loss.
    Returns (costs[B], d/d y_trans [B,T0,V], d/d y_pred [B,U+1,V])."""
    shift = 2  # util.py:100
    y_trans = np.asarray(y_trans, dtype=np.float64)
    y_pred = np.asarray(y_pred, dtype=np.float64)
    yt = y_trans[:, shift:, :]  # util.py:101
    in_len = np.asarray(input_length).reshape(-1) - shift  # util.py:102,110
    lab_len = np.asarray(label_length).reshape(-1)  # util.py:111
    acts = log_softmax(yt[:, :, None, :] + y_pred[:, None, :, :])  # util.py:107-109
    costs, g = rnnt_loss_and_grad(acts, labels, in_len, lab_len, blank_label, grad_scale)
    d_trans = np.zeros_like(y_trans)
    d_trans[:, shift:, :] = g.sum(axis=2)
    d_pred = g.sum(axis=1)
    return costs, d_trans, d_pred


def rnnt_lambda_func_v2(acts, labels, input_length, label_length, blank_label=39, grad_scale=None):
    """Float64 restatement of ``Util.rnnt_lambda_func_v2`` (reference
    util.py:53-74): the joint arrives already formed as [B,T0,U+1,V]."""
    shift = 2  # util.py:62
    acts = np.asarray(acts, dtype=np.float64)
    a = log_softmax(acts[:, shift:, :, :])  # util.py:63,66
    in_len = np.asarray(input_length).reshape(-1) - shift  # util.py:64,67
    lab_len = np.asarray(label_length).reshape(-1)  # util.py:68
    costs, g = rnnt_loss_and_grad(a, labels, in_len, lab_len, blank_label, grad_scale)
    d_acts = np.zeros_like(acts)
    d_acts[:, shift:, :, :] = g
    return costs, d_acts


def concatenate_lambda(y_trans, y_pred):
    """Float64 restatement of ``Util.concatenate_lambda`` (reference
    util.py:76-90): tile both outputs and concatenate -> [B,T,U+1,He+Hd]."""
    y_trans = np.asarray(y_trans, dtype=np.float64)
    y_pred = np.asarray(y_pred, dtype=np.float64)
    T, U1 = y_trans.shape[1], y_pred.shape[1]
    ex_t = np.tile(y_trans[:, :, None, :], (1, 1, U1, 1))  # util.py:83-84
    ex_p = np.tile(y_pred[:, None, :, :], (1, T, 1, 1))    # util.py:87-88
    return np.concatenate([ex_t, ex_p], axis=-1)            # util.py:90


def fc_joint_loss_and_grads(y_trans, y_pred, weights, labels, input_length, label_length, blank_label=39,
                            grad_scale=None):
    """Float64 restatement of the FC-joint model's loss head (reference
    keras_model.py:169-188): concatenate_lambda (util.py:76-90) -> get_fc
    (keras_model.py:83-104: Dense+tanh ... Dense linear, Keras kernels [in,out])
    -> rnnt_lambda_func_v2 (util.py:53-74), literally -- the tiled tensor IS
    built here.  Returns (costs[B], d y_trans, d y_pred, [(d kernel, d bias), ...])."""
    weights = [(np.asarray(k, np.float64), np.asarray(b, np.float64)) for k, b in weights]
    x0 = concatenate_lambda(y_trans, y_pred)
    xs = [x0]
    for li, (k, b) in enumerate(weights):
        z = xs[-1] @ k + b
        xs.append(z if li == len(weights) - 1 else np.tanh(z))  # keras_model.py:91-92
    costs, d = rnnt_lambda_func_v2(xs[-1], labels, input_length, label_length, blank_label, grad_scale)
    dws = [None] * len(weights)
    for li in range(len(weights) - 1, -1, -1):
        k, _ = weights[li]
        if li != len(weights) - 1:
            d = d * (1.0 - xs[li + 1] ** 2)
        xin = xs[li].reshape(-1, k.shape[0])
        d2 = d.reshape(-1, k.shape[1])
        dws[li] = (xin.T @ d2, d2.sum(axis=0))
        d = d @ k.T
    He = np.asarray(y_trans).shape[2]
    return costs, d[..., :He].sum(axis=2), d[..., He:].sum(axis=1), dws


def util_softmax(value, is_log=False, axis=None):
    """``Util.softmax`` (reference util.py:295-309) statement by statement, in
    float64.  With ``axis=0`` the reference recurses on every element of
    ``value`` WITHOUT ``is_log`` (util.py:299-301): rows of a matrix get a plain
    softmax each, and the elements of a vector -- scalars -- each become 1.0."""
    if axis is not None:
        assert axis == 0  # util.py:298
        return [util_softmax(elem) for elem in np.asarray(value, dtype=np.float64)]  # util.py:299-301
    value = np.asarray(value, dtype=np.float64)
    e_x = np.exp(value - np.max(value))  # util.py:303
    if is_log:
        with np.errstate(divide="ignore"):
            return np.log(e_x) - np.log(e_x.sum())  # util.py:305
    return e_x / e_x.sum()  # util.py:307


def decoder_step_logp(encoder_output_t, pred):
    """The decoder's inner step (reference rnnt_decoder.py:156-158, 200-202):
    out = pred + encoder_output_t; logp = Util.softmax(out, is_log=True)."""
    return util_softmax(np.asarray(pred, np.float64) + np.asarray(encoder_output_t, np.float64), is_log=True)


# ---------------------------------------------------------------------------
# Independent second opinions used to pin the oracle itself.
# ---------------------------------------------------------------------------

def brute_force_cost(logits, labels, T, U, blank):
    """-log of the sum over EVERY alignment path of its probability (pure
    Python, tiny lattices only).  A path is an interleaving of exactly U label
    emissions and T blanks whose last symbol is the blank leaving (T-1, U)."""
    lp = log_softmax(np.asarray(logits, dtype=np.float64)[:T, : U + 1, :])
    total = NEG_INF
    # choose positions of the U labels among the first T+U-1 moves; the final move is a blank
    for lab_pos in itertools.combinations(range(T + U - 1), U):
        lab_pos = set(lab_pos)
        t = u = 0
        s = 0.0
        ok = True
        for k in range(T + U - 1):
            if k in lab_pos:
                s += lp[t, u, labels[u]]
                u += 1
            else:
                s += lp[t, u, blank]
                t += 1
                if t >= T:
                    ok = False
                    break
        if not ok or t != T - 1 or u != U:
            continue
        s += lp[T - 1, U, blank]
        total = np.logaddexp(total, s)
    return -total


def numeric_grad(logits, labels, input_lengths, label_lengths, blank, eps=1e-6):
    """Central finite differences of sum_b cost_b w.r.t. every logit."""
    x = np.array(logits, dtype=np.float64)
    g = np.zeros_like(x)
    it = np.nditer(x, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        old = x[i]
        x[i] = old + eps
        cp = rnnt_loss_and_grad(x, labels, input_lengths, label_lengths, blank)[0].sum()
        x[i] = old - eps
        cm = rnnt_loss_and_grad(x, labels, input_lengths, label_lengths, blank)[0].sum()
        x[i] = old
        g[i] = (cp - cm) / (2 * eps)
    return g


# Public known-answer vector (warp-transducer / torchaudio unit test),
# re-derived independently with brute_force_cost + numeric_grad in
# tests/test_oracle.py before being trusted.
KNOWN_ANSWER = {
    "B": 1, "T": 2, "U": 2, "V": 5,
    "labels": [[1, 2]],
    "logits": [
        [[[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.6, 0.1, 0.1], [0.1, 0.1, 0.2, 0.8, 0.1]],
         [[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.2, 0.1, 0.1], [0.7, 0.1, 0.2, 0.1, 0.1]]]
    ],
    "blank0_cost": 4.495666774,
    "blank0_grad": [
        [[[-0.13116686, -0.3999268, 0.17703122, 0.17703122, 0.17703122],
          [-0.18572754, 0.12247054, -0.18168408, 0.12247054, 0.12247054],
          [-0.32091246, 0.06269139, 0.06928471, 0.12624497, 0.06269139]],
         [[0.05456068, -0.21824272, 0.05456068, 0.05456068, 0.05456068],
          [0.12073957, 0.12073957, -0.48295828, 0.12073957, 0.12073957],
          [-0.6925882, 0.16871117, 0.18645468, 0.16871117, 0.16871117]]]
    ],
    "blank4_cost": 5.095666774,
    "blank4_grad": [
        [[[0.17703122, -0.3999268, 0.17703122, 0.17703122, -0.13116686],
          [0.12247054, 0.12247054, -0.18168408, 0.12247054, -0.18572754],
          [0.0626914, 0.0626914, 0.06928471, 0.12624497, -0.32091246]],
         [[0.05456068, -0.21824272, 0.05456068, 0.05456068, 0.05456068],
          [0.12073957, 0.12073957, -0.48295828, 0.12073957, 0.12073957],
          [0.3074118, 0.16871117, 0.18645468, 0.16871117, -0.83128883]]]
    ],
}

# Second public known-answer vector: the batch case of the warp-transducer
# binding's own unit test (B=2, T=4, U=2, V=3, blank 0; also carried by
# torchaudio's transducer-loss tests).  Costs and gradient are the PUBLISHED
# numbers (six decimals, computed upstream in float32), not outputs of this
# module; tests/test_oracle.py checks the oracle against them.
KNOWN_ANSWER_BATCH = {
    "B": 2, "T": 4, "U": 2, "V": 3, "blank": 0,
    "labels": [[1, 2], [1, 1]],
    "input_lengths": [4, 4],
    "label_lengths": [2, 2],
    "logits": [
        0.065357, 0.787530, 0.081592, 0.529716, 0.750675, 0.754135, 0.609764, 0.868140, 0.622532,
        0.668522, 0.858039, 0.164539, 0.989780, 0.944298, 0.603168, 0.946783, 0.666203, 0.286882,
        0.094184, 0.366674, 0.736168, 0.166680, 0.714154, 0.399400, 0.535982, 0.291821, 0.612642,
        0.324241, 0.800764, 0.524106, 0.779195, 0.183314, 0.113745, 0.240222, 0.339470, 0.134160,
        0.505562, 0.051597, 0.640290, 0.430733, 0.829473, 0.177467, 0.320700, 0.042883, 0.302803,
        0.675178, 0.569537, 0.558474, 0.083132, 0.060165, 0.107958, 0.748615, 0.943918, 0.486356,
        0.418199, 0.652408, 0.024243, 0.134582, 0.366342, 0.295830, 0.923670, 0.689929, 0.741898,
        0.250005, 0.603430, 0.987289, 0.592606, 0.884672, 0.543450, 0.660770, 0.377128, 0.358021,
    ],  # row-major [B, T, U+1, V]
    "costs": [4.2806528590890736, 3.9384369822503591],
    "grad": [
        -0.186844, -0.062555, 0.249399, -0.203377, 0.202399, 0.000977, -0.141016, 0.079123, 0.061893,
        -0.011552, -0.081280, 0.092832, -0.154257, 0.229433, -0.075176, -0.246593, 0.146405, 0.100188,
        -0.012918, -0.061593, 0.074512, -0.055986, 0.219831, -0.163845, -0.497627, 0.209240, 0.288387,
        0.013605, -0.030220, 0.016615, 0.113925, 0.062781, -0.176706, -0.667078, 0.367659, 0.299419,
        -0.356344, -0.055347, 0.411691, -0.096922, 0.029459, 0.067463, -0.063518, 0.027654, 0.035863,
        -0.154499, -0.073942, 0.228441, -0.166790, -0.000088, 0.166878, -0.172370, 0.105565, 0.066804,
        0.023875, -0.118256, 0.094381, -0.104707, -0.108934, 0.213642, -0.369844, 0.180118, 0.189726,
        0.025714, -0.079462, 0.053748, 0.122328, -0.238789, 0.116460, -0.598687, 0.302203, 0.296484,
    ],  # row-major [B, T, U+1, V]
}


def known_answer_batch():
    """KNOWN_ANSWER_BATCH as arrays: (logits f64 [2,4,3,3], labels, input_lengths,
    label_lengths, blank, costs f64 [2], grad f64 [2,4,3,3])."""
    K = KNOWN_ANSWER_BATCH
    shape = (K["B"], K["T"], K["U"] + 1, K["V"])
    return (np.array(K["logits"], np.float64).reshape(shape), np.array(K["labels"], np.int32),
            np.array(K["input_lengths"], np.int32), np.array(K["label_lengths"], np.int32), K["blank"],
            np.array(K["costs"], np.float64), np.array(K["grad"], np.float64).reshape(shape))
