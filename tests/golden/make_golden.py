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


def main():
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
