"""The log-softmax of the reference decoder's inner step, batched on the MI355X.

``KerasRNNTDecoder.beam_search`` (reference rnnt/rnnt_decoder.py:95-233) forms
``out = pred + encoder_output_t`` and ``logp = Util.softmax(out, is_log=True)``
for one hypothesis and one frame at a time in numpy (rnnt_decoder.py:156-158,
200-202).  ``joint_log_softmax`` evaluates any number of such pairs in one
kernel launch; ``softmax`` keeps the name, arguments and behaviour of
``Util.softmax`` (reference rnnt/base/util.py:295-309) for device tensors.
"""
from __future__ import annotations

import torch

from . import _lib
from .loss import _ptr, _stream_ptr


_UNUSED_FLAG = {}  # per device: the flag word handed to the kernel when it cannot be set (indices checked on the host)


def joint_log_softmax(enc, pred=None, enc_index=None, pred_index=None):
    """log_softmax(enc[enc_index[n]] + pred[pred_index[n]]) over the last axis.

    enc: [Ne, V] float32 encoder rows; pred: [Np, V] prediction-network rows or
    None (then the log-softmax of the selected enc rows); the index tensors are
    int [N] or None (row n of the respective tensor).  Returns [N, V]."""
    if enc.device.type != "cuda":
        raise ValueError("joint_log_softmax: tensors must live on an MI355X device (there is no CPU path)")
    if enc.dim() != 2 or (pred is not None and (pred.dim() != 2 or pred.shape[1] != enc.shape[1])):
        raise ValueError("joint_log_softmax: expected enc[Ne,V] and pred[Np,V]")
    dev = enc.device
    enc = enc.float().contiguous()
    pred = None if pred is None else pred.to(dev).float().contiguous()
    enc_index = None if enc_index is None else torch.as_tensor(enc_index)
    pred_index = None if pred_index is None else torch.as_tensor(pred_index)
    fi = None if enc_index is None else enc_index.to(dev, torch.int32).contiguous()
    gi = None if pred_index is None else pred_index.to(dev, torch.int32).contiguous()
    if fi is not None and gi is not None and fi.numel() != gi.numel():
        raise ValueError("joint_log_softmax: enc_index and pred_index differ in length")
    if fi is not None:
        n = fi.numel()
    elif gi is not None:
        n = gi.numel()
    elif pred is not None:
        if pred.shape[0] != enc.shape[0]:
            raise ValueError("joint_log_softmax: without index tensors enc and pred need the same number of rows")
        n = enc.shape[0]
    else:
        n = enc.shape[0]
    V = enc.shape[1]
    out = torch.empty((n, V), dtype=torch.float32, device=dev)
    if n == 0:
        return out
    # Row indices given as host tensors are checked here; without index tensors there is nothing to check.  Only
    # indices that already live on the device are checked by the kernel, and only then is its flag read back (a
    # stream sync, ~30 us): the decoders call this once per frame and hypothesis set.
    on_device = False
    for idx, rows, what in ((enc_index, enc.shape[0], "enc"), (pred_index, 0 if pred is None else pred.shape[0], "pred")):
        if idx is None:
            continue
        if idx.is_cuda:
            on_device = True
        elif idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= rows):
            raise IndexError(f"joint_log_softmax: a row index lies outside {what}")
    if on_device:
        bad = torch.zeros(1, dtype=torch.int32, device=dev)
    else:
        bad = _UNUSED_FLAG.get(dev)
        if bad is None:
            bad = _UNUSED_FLAG[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().rnnt_mi355x_joint_log_softmax(_ptr(enc), enc.shape[0], _ptr(pred), 0 if pred is None else pred.shape[0],
                                                      _ptr(fi), _ptr(gi), n, V, _ptr(out), _ptr(bad), _stream_ptr(dev))
    _lib.check(rc, "rnnt_mi355x_joint_log_softmax")
    if on_device and int(bad.item()):
        raise IndexError("joint_log_softmax: a row index lies outside its tensor")
    return out


def softmax(value, is_log=False, axis=None):
    """``Util.softmax`` (reference util.py:295-309) for a device tensor.

    axis=None: exp(value - max(value)) over ALL elements, normalised (is_log:
    its logarithm).  axis=0 reproduces what the reference does there: every
    element of ``value`` along axis 0 gets its own, non-logarithmic softmax and
    ``is_log`` is ignored (util.py:297-301) -- for a 1-D vector that is a list of
    ones (each "element" is a scalar), as at rnnt_decoder.py:171.  Any other
    axis fails, like the reference's assertion."""
    if axis is not None:
        assert axis == 0
        if value.dim() == 1:
            return torch.ones_like(value, dtype=torch.float32)
        rows = value.reshape(value.shape[0], -1)
        return joint_log_softmax(rows).exp().reshape(value.shape)
    logp = joint_log_softmax(value.reshape(1, -1)).reshape(value.shape)
    return logp if is_log else logp.exp()
