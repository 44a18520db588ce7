"""The two loss lambdas of the reference model, re-hosted on the MI355X loss.

``Util.rnnt_lambda_func`` (reference rnnt/base/util.py:93-117) and
``Util.rnnt_lambda_func_v2`` (reference rnnt/base/util.py:53-74) are what the
Keras graph wraps in ``Lambda`` layers (reference rnnt/keras_model.py:165-168,
186-188).  Same names, same argument lists, same crop / length handling /
hard-coded default blank, same ``[B]`` result.  The explicit
``tf.nn.log_softmax`` of util.py:66/109 is not executed as a separate pass:
the loss kernels normalise each row themselves, which yields the same costs
and, through autograd, the same gradients with respect to the lambda inputs.
"""
from __future__ import annotations

from .loss import rnnt_joint_loss, rnnt_loss

SHIFT = 2  # "the first couple outputs of the RNN tend to be garbage" (util.py:60-62, 98-100)
REFERENCE_BLANK_LABEL = 39  # util.py:72, util.py:115


def rnnt_lambda_func(args, blank_label=REFERENCE_BLANK_LABEL, fused=True):
    """[y_trans[B,T0,V], y_pred[B,U+1,V], labels[B,U], input_length[B,1], label_length[B,1]] -> costs[B]
    (reference util.py:93-117).  fused=True (default) never forms the
    [B,T,U+1,V] joint of util.py:107-109; fused=False builds it with a broadcast
    add and calls rnnt_loss on it, exactly as the reference does."""
    y_trans, y_pred, labels, input_length, label_length = args
    batch_size = y_trans.shape[0]                        # util.py:106
    input_length = input_length - SHIFT                  # util.py:102 (no in-place edit of the caller's tensor)
    input_length = input_length.reshape(batch_size)      # util.py:110
    label_length = label_length.reshape(batch_size)      # util.py:111
    if fused and y_trans.shape[2] <= 4096:
        costs = rnnt_joint_loss(y_trans, y_pred, labels, input_length, label_length, blank_label,
                                t_offset=SHIFT)          # util.py:101, 107-109, 113-115
        return costs.reshape(batch_size)                 # util.py:117
    y_trans = y_trans[:, SHIFT:, :]                      # util.py:101
    acts = y_trans.unsqueeze(2) + y_pred.unsqueeze(1)    # util.py:107-109 (BT1V + B1UV)
    # acts is a fresh temporary nobody else reads: let backward reuse its storage
    costs = rnnt_loss(acts, labels, input_length, label_length, blank_label, inplace=True)  # util.py:113-115
    return costs.reshape(batch_size)                     # util.py:117


def rnnt_lambda_func_v2(args, blank_label=REFERENCE_BLANK_LABEL):
    """[acts[B,T0,U+1,V], labels[B,U], input_length[B,1], label_length[B,1]] -> costs[B]
    (reference util.py:53-74); acts comes from the FC joint.  The crop
    ``acts[:, 2:, :, :]`` (util.py:63) is not materialised: the loss is told to
    start at frame 2 (a strided view of the same storage)."""
    acts, labels, input_length, label_length = args
    batch_size = acts.shape[0]                           # util.py:58
    input_length = input_length - SHIFT                  # util.py:64
    input_length = input_length.reshape(batch_size)      # util.py:67
    label_length = label_length.reshape(batch_size)      # util.py:68
    costs = rnnt_loss(acts, labels, input_length, label_length, blank_label, t_offset=SHIFT)  # util.py:63, 70-72
    return costs.reshape(batch_size)                     # util.py:74
