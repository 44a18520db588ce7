"""Generates tests/golden/*.npz from the float64 oracle (oracle/rnnt_oracle.py).

The reference holds no fixture for the loss path and none of its loss code can
run here (its arithmetic lives in the absent warprnnt_tensorflow package), so
these vectors are produced by the oracle after it has been pinned by the public
known-answer vector, brute-force path enumeration and finite differences
(tests/test_oracle.py).  Inputs are stored as float32 (what the HIP path
consumes); expected outputs are float64 computed from those float32 values.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rnnt_oracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    # name, B, T, U, V, blank, ragged, logit scale
    ("ragged_small_midblank", 3, 7, 4, 11, 5, True, 1.0),
    ("timit62_blank39", 2, 12, 6, 62, 39, True, 1.0),
    ("timit40_blank39", 2, 9, 5, 40, 39, True, 2.0),
    ("no_labels_u0", 2, 5, 0, 8, 0, True, 1.0),
    ("single_frame_t1", 2, 1, 3, 9, 8, False, 1.0),
    ("odd_vocab_63", 2, 6, 3, 63, 62, True, 3.0),
    ("wide_vocab_1024", 1, 5, 3, 1024, 1023, False, 1.0),
    ("peaked_logits", 2, 10, 4, 16, 0, True, 12.0),
]


def make_case(name, B, T, U, V, blank, ragged, scale, seed):
    rng = np.random.default_rng(seed)
    logits = (rng.standard_normal((B, T, U + 1, V)) * scale).astype(np.float32)
    labels = rng.integers(0, V - 1, size=(B, U)).astype(np.int32)
    labels[labels >= blank] += 1  # uniform over [0,V) \ {blank}
    if ragged and B > 1:
        in_len = rng.integers(max(1, T // 2), T + 1, size=B).astype(np.int32)
        lab_len = rng.integers(0, U + 1, size=B).astype(np.int32)
        in_len[0], lab_len[0] = T, U
    else:
        in_len = np.full(B, T, np.int32)
        lab_len = np.full(B, U, np.int32)
    grad_scale = rng.uniform(0.25, 1.5, size=B).astype(np.float32)
    costs, grad = orc.rnnt_loss_and_grad(logits, labels, in_len, lab_len, blank, grad_scale)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), logits=logits, labels=labels, in_len=in_len,
                        lab_len=lab_len, blank=np.int32(blank), grad_scale=grad_scale, costs=costs, grad=grad)


def make_op_fixtures():
    """Fixtures of the callers either side of the loss: the additive-joint lambda, the FC-joint loss head and
    the decoder's inner step (tests/golden/ops/)."""
    ops = os.path.join(HERE, "ops")
    os.makedirs(ops, exist_ok=True)
    rng = np.random.default_rng(777)
    # Util.rnnt_lambda_func: y_trans [B,T0,V], y_pred [B,U+1,V], lengths [B,1], blank 39, crop of 2 frames
    B, T0, U, V = 3, 12, 5, 62
    yt = rng.standard_normal((B, T0, V)).astype(np.float32)
    yp = rng.standard_normal((B, U + 1, V)).astype(np.float32)
    lab = rng.integers(0, 39, (B, U)).astype(np.int32)
    il = np.array([[12], [9], [11]], np.int32)
    ll = np.array([[5], [3], [4]], np.int32)
    gs = np.full(B, 1.0 / B)
    costs, dt, dp = orc.rnnt_lambda_func(yt, yp, lab, il, ll, blank_label=39, grad_scale=gs)
    np.savez_compressed(os.path.join(ops, "lambda_additive_joint.npz"), y_trans=yt, y_pred=yp, labels=lab,
                        input_length=il, label_length=ll, grad_scale=gs, costs=costs, d_trans=dt, d_pred=dp)
    # FC-joint loss head: concatenate_lambda -> Dense(tanh) -> Dense -> rnnt_lambda_func_v2
    He, Hd, H = 10, 6, 8
    yt2 = (rng.standard_normal((B, T0, He)) * 0.5).astype(np.float32)
    yp2 = (rng.standard_normal((B, U + 1, Hd)) * 0.5).astype(np.float32)
    k1 = rng.uniform(-0.5, 0.5, (He + Hd, H)).astype(np.float32)
    b1 = rng.uniform(-0.1, 0.1, H).astype(np.float32)
    k2 = rng.uniform(-0.5, 0.5, (H, V)).astype(np.float32)
    b2 = rng.uniform(-0.1, 0.1, V).astype(np.float32)
    c2, d_t, d_p, dws = orc.fc_joint_loss_and_grads(yt2, yp2, [(k1, b1), (k2, b2)], lab, il, ll, blank_label=39,
                                                    grad_scale=gs)
    np.savez_compressed(os.path.join(ops, "fc_joint_two_layers.npz"), y_trans=yt2, y_pred=yp2, k1=k1, b1=b1, k2=k2,
                        b2=b2, labels=lab, input_length=il, label_length=ll, grad_scale=gs, costs=c2, d_trans=d_t,
                        d_pred=d_p, dk1=dws[0][0], db1=dws[0][1], dk2=dws[1][0], db2=dws[1][1])
    # decoder inner step for a small beam
    enc = (rng.standard_normal((4, V)) * 3).astype(np.float32)
    pred = (rng.standard_normal((6, V)) * 3).astype(np.float32)
    fi = np.array([0, 0, 1, 1, 2, 3, 3, 3, 2], np.int32)
    gi = np.array([0, 1, 0, 5, 2, 3, 4, 0, 5], np.int32)
    logp = np.stack([orc.decoder_step_logp(enc[i], pred[j]) for i, j in zip(fi, gi)])
    np.savez_compressed(os.path.join(ops, "decoder_step.npz"), enc=enc, pred=pred, enc_index=fi, pred_index=gi, logp=logp)
    # the second public known-answer vector, published values
    x, lab2, il2, ll2, blank2, pc, pg = orc.known_answer_batch()
    np.savez_compressed(os.path.join(ops, "known_answer_batch.npz"), logits=x.astype(np.float32), labels=lab2,
                        in_len=il2, lab_len=ll2, blank=np.int32(blank2), published_costs=pc, published_grad=pg)
    print("wrote 4 fixtures to", ops)


def main():
    make_op_fixtures()
    for k, c in enumerate(CASES):
        make_case(*c, seed=20240 + k)
    K = orc.KNOWN_ANSWER
    np.savez_compressed(
        os.path.join(HERE, "known_answer.npz"), logits=np.array(K["logits"], np.float32),
        labels=np.array(K["labels"], np.int32), in_len=np.array([K["T"]], np.int32),
        lab_len=np.array([K["U"]], np.int32), blank0_cost=np.float64(K["blank0_cost"]),
        blank0_grad=np.array(K["blank0_grad"]), blank4_cost=np.float64(K["blank4_cost"]),
        blank4_grad=np.array(K["blank4_grad"]))
    print("wrote", len(CASES) + 1, "fixtures to", HERE)


if __name__ == "__main__":
    main()
