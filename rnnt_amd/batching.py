"""Host-side batch layout of the loss inputs: what the reference's batch
producer hands to the model (reference rnnt/base/data_generator.py:391-426),
so that callers of the reference find the same arrays under the same keys.

Shapes only -- features and label sequences come from the caller; nothing here
reads audio or text.
"""
from __future__ import annotations

import numpy as np

# keys of the reference's input / output dictionaries (reference rnnt/base/common.py:34-38,46)
INPUT_TRANS = "INPUT_TRANS"
INPUT_PREDS = "INPUT_PREDS"
INPUT_LABEL = "INPUT_LABEL"
INPUT_INLEN = "INPUT_INLEN"
INPUT_LBLEN = "INPUT_LBLEN"
LOSS_RNNT = "LOSS_RNNT"


def pad_batch(features, label_seqs, vocab_size):
    """Pads one batch the way reference data_generator.py:391-426 does.

    features: B arrays [T_b, F]; label_seqs: B integer sequences of length U_b
    with entries in [0, vocab_size).  Returns (inputs, outputs):

      inputs[INPUT_TRANS]  [B, max T, F]           features, zero padded
      inputs[INPUT_PREDS]  [B, max U + 1, vocab]   one-hot of the labels shifted by one position: row 0 (the
                                                   prepended blank) and the padding rows are all zero
      inputs[INPUT_LABEL]  [B, max U]              labels, padded with 0
      inputs[INPUT_INLEN]  [B]                     T_b   (float64 like the reference's; the model's int32 Inputs
      inputs[INPUT_LBLEN]  [B]                     U_b    cast them, keras_model.py:125-127)
      outputs[LOSS_RNNT]   [B] zeros               the dummy target of the identity loss (keras_model.py:215-217)

    The lengths count frames BEFORE the model's crop of two frames (util.py:62-64): rnnt_lambda_func /
    rnnt_lambda_func_v2 subtract it.
    """
    B = len(features)
    if B == 0 or len(label_seqs) != B:
        raise ValueError("features and label_seqs must hold the same, non-zero number of utterances")
    feats = [np.asarray(f) for f in features]
    if any(f.ndim != 2 or f.shape[1] != feats[0].shape[1] or f.shape[0] == 0 for f in feats):
        raise ValueError("every feature array must be [T_b >= 1, F] with one F")
    seqs = [np.asarray(s, dtype=np.int64).reshape(-1) for s in label_seqs]
    if any(s.size and (s.min() < 0 or s.max() >= vocab_size) for s in seqs):
        raise ValueError("a label is outside [0, vocab_size)")
    max_t = max(f.shape[0] for f in feats)
    max_u = max(s.size for s in seqs)
    input_tran = np.zeros([B, max_t, feats[0].shape[1]])
    input_pred = np.zeros([B, max_u + 1, vocab_size])
    label_rnnt = np.zeros([B, max_u])
    input_length = np.zeros([B])
    label_length = np.zeros([B])
    for i, (f, s) in enumerate(zip(feats, seqs)):
        input_length[i] = f.shape[0]
        input_tran[i, : f.shape[0]] = f
        label_length[i] = s.size
        label_rnnt[i, : s.size] = s
        input_pred[i, np.arange(1, s.size + 1), s] = 1
    inputs = {INPUT_TRANS: input_tran, INPUT_PREDS: input_pred, INPUT_INLEN: input_length, INPUT_LBLEN: label_length,
              INPUT_LABEL: label_rnnt}
    return inputs, {LOSS_RNNT: np.zeros([B])}
