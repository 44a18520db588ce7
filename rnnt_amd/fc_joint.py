"""Front end of the reference's fully-connected joint model (ModelType.RNNT_FF).

The reference wires (rnnt/keras_model.py:169-188)

    joint_pred = Lambda(Util.concatenate_lambda)([encoder.output, decoder.output])   # util.py:76-90
    acts       = KerasModel.get_fc(joint_pred, V, hidden, layers, dropout)           # keras_model.py:83-104
    loss_out   = Lambda(Util.rnnt_lambda_func_v2)([acts, label, input_length, label_length])

``concatenate_lambda`` tiles both network outputs into [B,T0,U+1,He+Hd] and the
first Dense layer then multiplies that tiled tensor.  The first layer is linear
in the concatenation, so here it is computed as two small matrix products
(library GEMMs) on the un-tiled outputs followed by one HIP kernel that
broadcasts, adds and applies tanh; backwards one HIP pass over the upstream
gradient gives both reduced gradients.  With a single layer the joint is an
additive one and goes straight to the fused additive-joint loss, so not even
[B,T,U+1,V] is formed.  The following layers multiply a [B*T*(U+1), hidden]
matrix by a small kernel: up to 128 x 128 they run on hand-written tall-skinny
MFMA kernels (weights resident in LDS), beyond that on the library GEMM.  Dropout (training-time noise, keras_model.py:102-103)
is not part of these functions.
"""
from __future__ import annotations

import torch

from . import _lib
from .lambdas import REFERENCE_BLANK_LABEL, SHIFT, rnnt_lambda_func
from .loss import _ptr, _stream_ptr, rnnt_loss


def concatenate_lambda(args):
    """[y_trans[B,T,He], y_pred[B,U+1,Hd]] -> [B,T,U+1,He+Hd], the tiled
    concatenation of reference util.py:76-90.  Kept for callers that want the
    tensor itself; fc_joint / fc_joint_loss never build it."""
    y_trans, y_pred = args
    T, U1 = y_trans.shape[1], y_pred.shape[1]
    ex_y_trans = y_trans.unsqueeze(2).expand(-1, -1, U1, -1)   # util.py:83-84
    ex_y_pred = y_pred.unsqueeze(1).expand(-1, T, -1, -1)      # util.py:87-88
    return torch.cat([ex_y_trans, ex_y_pred], dim=-1)          # util.py:90


class _FirstLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pe, pd):
        B, T, H = pe.shape
        U1 = pd.shape[1]
        h = torch.empty((B, T, U1, H), dtype=torch.float32, device=pe.device)
        with torch.cuda.device(pe.device):
            rc = _lib.lib().rnnt_mi355x_fc_joint_forward(_ptr(pe), _ptr(pd), B, T, U1, H, _ptr(h), _stream_ptr(pe.device))
        _lib.check(rc, "rnnt_mi355x_fc_joint_forward")
        ctx.save_for_backward(pe, pd)
        return h

    @staticmethod
    def backward(ctx, dh):
        pe, pd = ctx.saved_tensors
        B, T, H = pe.shape
        U1 = pd.shape[1]
        dh = dh.contiguous()
        dpe, dpd = torch.empty_like(pe), torch.empty_like(pd)
        with torch.cuda.device(pe.device):
            ws = torch.empty(_lib.fc_joint_workspace_bytes(B, T, U1, H), dtype=torch.uint8, device=pe.device)
            rc = _lib.lib().rnnt_mi355x_fc_joint_backward(_ptr(pe), _ptr(pd), _ptr(dh), B, T, U1, H, _ptr(dpe), _ptr(dpd),
                                                          _ptr(ws), _stream_ptr(pe.device))
        _lib.check(rc, "rnnt_mi355x_fc_joint_backward")
        return dpe, dpd


def fc_joint_first_layer(pe, pd):
    """tanh(pe[B,T,1,H] + pd[B,1,U+1,H]) -> [B,T,U+1,H] (differentiable)."""
    if pe.device.type != "cuda" or pd.device != pe.device:
        raise ValueError("fc_joint_first_layer: tensors must live on one MI355X device (there is no CPU path)")
    if pe.dim() != 3 or pd.dim() != 3 or pe.shape[0] != pd.shape[0] or pe.shape[2] != pd.shape[2]:
        raise ValueError(f"fc_joint_first_layer: expected pe[B,T,H], pd[B,U+1,H]; got {tuple(pe.shape)}, {tuple(pd.shape)}")
    return _FirstLayerFn.apply(pe.float().contiguous(), pd.float().contiguous())


DENSE_MAX_DIM = 128  # rnnt_mi355x_dense_*: both layer widths at most this, else the library GEMM


class _DenseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kernel, bias, act):
        K, N = kernel.shape
        M = x.numel() // K
        y = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = _lib.lib().rnnt_mi355x_dense_forward(_ptr(x), _ptr(kernel), _ptr(bias), M, K, N, act, _ptr(y),
                                                     _stream_ptr(x.device))
        _lib.check(rc, "rnnt_mi355x_dense_forward")
        ctx.save_for_backward(x, kernel, y if act == 1 else None)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        x, kernel, y = ctx.saved_tensors
        K, N = kernel.shape
        M = x.numel() // K
        dy = dy.contiguous()
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty_like(x) if need_dx else None
        dw, db = torch.empty_like(kernel), torch.empty(N, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            ws = torch.empty(_lib.dense_workspace_bytes(M, K, N), dtype=torch.uint8, device=x.device)
            rc = _lib.lib().rnnt_mi355x_dense_backward(_ptr(x), _ptr(kernel), _ptr(y), _ptr(dy), M, K, N, ctx.act,
                                                      _ptr(dx), _ptr(dw), _ptr(db), _ptr(ws), _stream_ptr(x.device))
        _lib.check(rc, "rnnt_mi355x_dense_backward")
        return dx, dw, db, None


def joint_dense(x, kernel, bias, activation="linear"):
    """One TimeDistributed(Dense) of get_fc (keras_model.py:88-101) over x[..., K]:
    activation(x @ kernel[K,N] + bias[N]), "linear" or "tanh".  Widths up to
    DENSE_MAX_DIM run on the hand-written tall-skinny kernels; wider layers go
    to the library GEMM, which is efficient there."""
    if activation not in ("linear", "tanh"):
        raise ValueError("joint_dense: activation must be 'linear' or 'tanh'")
    K, N = kernel.shape
    if x.shape[-1] != K:
        raise ValueError(f"joint_dense: x has {x.shape[-1]} features, kernel expects {K}")
    for name, t in (("x", x), ("kernel", kernel), ("bias", bias)):
        if t.device.type != "cuda":
            raise ValueError(f"joint_dense: {name} must live on an MI355X device (there is no CPU path)")
    if K > DENSE_MAX_DIM or N > DENSE_MAX_DIM:  # wide layers: the vendor GEMM (rocBLAS / hipBLASLt through torch)
        z = torch.matmul(x, kernel) + bias
        return torch.tanh(z) if activation == "tanh" else z
    return _DenseFn.apply(x.float().contiguous(), kernel.float().contiguous(), bias.float().contiguous(),
                          1 if activation == "tanh" else 0)


def fc_joint(y_trans, y_pred, weights):
    """concatenate_lambda + get_fc without the tiled tensor.

    weights: [(kernel[in,out], bias[out]), ...] in Keras Dense convention, one
    pair per layer (keras_model.py:87-101): tanh on all layers but the last,
    which is linear and has V outputs.  The first kernel has He+Hd rows.
    Returns acts[B,T,U+1,V]."""
    if len(weights) < 1:
        raise ValueError("fc_joint: at least one Dense layer")
    He = y_trans.shape[2]
    k1, b1 = weights[0]
    if k1.shape[0] != He + y_pred.shape[2]:
        raise ValueError(f"fc_joint: first kernel has {k1.shape[0]} rows, expected {He} + {y_pred.shape[2]}")
    pe = torch.matmul(y_trans, k1[:He])             # [B,T,H]
    pd = torch.matmul(y_pred, k1[He:]) + b1         # [B,U+1,H]
    if len(weights) == 1:                           # linear single layer: an additive joint
        return pe.unsqueeze(2) + pd.unsqueeze(1)
    h = fc_joint_first_layer(pe, pd)
    for k, b in weights[1:-1]:
        h = joint_dense(h, k, b, "tanh")
    k, b = weights[-1]
    return joint_dense(h, k, b, "linear")


def fc_joint_loss(y_trans, y_pred, weights, labels, input_length, label_length, blank_label=REFERENCE_BLANK_LABEL):
    """The loss of the FC-joint model from the two network outputs:
    rnnt_lambda_func_v2([get_fc(concatenate_lambda([y_trans, y_pred])), labels,
    input_length, label_length]) (keras_model.py:169-188) -> costs[B].
    The first SHIFT frames are dropped before the joint instead of after it
    (every layer acts per position, so the result is the same)."""
    B = y_trans.shape[0]
    if len(weights) == 1:
        He = y_trans.shape[2]
        k1, b1 = weights[0]
        f = torch.matmul(y_trans, k1[:He])
        g = torch.matmul(y_pred, k1[He:]) + b1
        return rnnt_lambda_func([f, g, labels, input_length, label_length], blank_label=blank_label, fused=True)
    acts = fc_joint(y_trans[:, SHIFT:], y_pred, weights)
    il = (input_length - SHIFT).reshape(B)
    ll = label_length.reshape(B)
    return rnnt_loss(acts, labels, il, ll, blank_label, inplace=True).reshape(B)
