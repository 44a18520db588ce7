"""FC-joint front end (concatenate_lambda + get_fc + rnnt_lambda_func_v2 of the
reference's RNNT_FF model) on the MI355X against the float64 oracle, which builds
the tiled tensor literally.  Bars: cost 1e-5 relative, gradients 1e-4 absolute."""
import ctypes

import numpy as np
import pytest
import torch

import rnnt_amd
from oracle import rnnt_oracle as orc
from rnnt_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
COST_RTOL = 1e-5
GRAD_ATOL = 1e-4


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


# B, T, U1, H: vector and scalar paths, several LDS tiles per utterance (U1*H > 8192), ragged frame chunks,
# an H that needs more than the default dynamic LDS
FIRST_LAYER_SHAPES = [(2, 9, 5, 100), (1, 8, 3, 7), (2, 17, 40, 256), (1, 5, 101, 100), (1, 3, 4, 1024), (2, 1, 1, 1),
                      (2, 3, 70, 62), (1, 2, 300, 7)]  # the forward cut by position in memory: 8-byte and scalar paths


@pytest.mark.parametrize("B,T,U1,H", FIRST_LAYER_SHAPES)
def test_first_layer_through_the_c_abi(B, T, U1, H):
    rng = np.random.default_rng(B + 3 * T + 5 * U1 + 7 * H)
    pe = rng.standard_normal((B, T, H)).astype(np.float32)
    pd = rng.standard_normal((B, U1, H)).astype(np.float32)
    dh = rng.standard_normal((B, T, U1, H)).astype(np.float32)
    L = _lib.lib()
    tpe, tpd, tdh = (torch.from_numpy(a).to(DEV) for a in (pe, pd, dh))
    h = torch.empty((B, T, U1, H), device=DEV)
    assert L.rnnt_mi355x_fc_joint_forward(_p(tpe), _p(tpd), B, T, U1, H, _p(h), None) == 0
    ref = np.tanh(pe.astype(np.float64)[:, :, None, :] + pd.astype(np.float64)[:, None, :, :])
    assert np.abs(h.cpu().numpy() - ref).max() < 2e-6
    ws = torch.empty(_lib.fc_joint_workspace_bytes(B, T, U1, H), dtype=torch.uint8, device=DEV)
    dpe, dpd = torch.full_like(tpe, 7.0), torch.full_like(tpd, 7.0)
    assert L.rnnt_mi355x_fc_joint_backward(_p(tpe), _p(tpd), _p(tdh), B, T, U1, H, _p(dpe), _p(dpd), _p(ws), None) == 0
    d = dh.astype(np.float64) * (1.0 - ref ** 2)
    assert np.abs(dpe.cpu().numpy() - d.sum(2)).max() < 1e-4 * max(1.0, np.sqrt(U1))
    assert np.abs(dpd.cpu().numpy() - d.sum(1)).max() < 1e-4 * max(1.0, np.sqrt(T))
    # bit-reproducible
    dpe2, dpd2 = torch.empty_like(tpe), torch.empty_like(tpd)
    assert L.rnnt_mi355x_fc_joint_backward(_p(tpe), _p(tpd), _p(tdh), B, T, U1, H, _p(dpe2), _p(dpd2), _p(ws), None) == 0
    assert torch.equal(dpe, dpe2) and torch.equal(dpd, dpd2)


def test_first_layer_rejects_bad_arguments():
    L = _lib.lib()
    x = torch.zeros(8, device=DEV)
    assert L.rnnt_mi355x_fc_joint_forward(_p(x), _p(x), 1, 1, 1, 4096, _p(x), None) == 2      # H beyond the limit
    assert L.rnnt_mi355x_fc_joint_forward(None, _p(x), 1, 1, 1, 4, _p(x), None) == 1
    assert L.rnnt_mi355x_fc_joint_forward(_p(x), _p(x), 0, 1, 1, 4, _p(x), None) == 1
    with pytest.raises(ValueError):
        rnnt_amd.fc_joint_first_layer(torch.zeros(1, 2, 4), torch.zeros(1, 2, 4))               # CPU tensors


def test_concatenate_lambda_matches_reference_layout():
    rng = np.random.default_rng(3)
    yt, yp = rng.standard_normal((2, 5, 3)).astype(np.float32), rng.standard_normal((2, 4, 2)).astype(np.float32)
    out = rnnt_amd.concatenate_lambda([torch.from_numpy(yt).to(DEV), torch.from_numpy(yp).to(DEV)])
    np.testing.assert_array_equal(out.cpu().numpy(), orc.concatenate_lambda(yt, yp).astype(np.float32))


def _weights(rng, dims):
    return [(rng.uniform(-0.5, 0.5, (dims[i], dims[i + 1])).astype(np.float32),
             rng.uniform(-0.1, 0.1, dims[i + 1]).astype(np.float32)) for i in range(len(dims) - 1)]


# layers: the reference's sample (timit-joint-rnnt.conf: enc 128 bi -> 256, dec 128, joint 100 x 2 layers, V=62),
# a single linear layer (= additive joint), three layers
@pytest.mark.parametrize("He,Hd,hidden,V,blank", [(256, 128, [100], 62, 39), (12, 6, [], 40, 39), (10, 5, [16, 12], 30, 0)])
def test_fc_joint_loss_matches_oracle(He, Hd, hidden, V, blank):
    rng = np.random.default_rng(He + len(hidden))
    B, T0, U = 3, 13, 5
    yt = rng.standard_normal((B, T0, He)).astype(np.float32) * 0.5
    yp = rng.standard_normal((B, U + 1, Hd)).astype(np.float32) * 0.5
    ws = _weights(rng, [He + Hd] + hidden + [V])
    lab = rng.integers(0, V - 1, (B, U)).astype(np.int32)
    lab[lab >= blank] += 1
    il = np.array([[13], [9], [11]], np.int32)
    ll = np.array([[5], [2], [4]], np.int32)
    gs = np.full(B, 1.0 / B)
    co, dt, dp, dws = orc.fc_joint_loss_and_grads(yt, yp, ws, lab, il, ll, blank_label=blank, grad_scale=gs)

    t_yt = torch.from_numpy(yt).to(DEV).requires_grad_(True)
    t_yp = torch.from_numpy(yp).to(DEV).requires_grad_(True)
    t_ws = [(torch.from_numpy(k).to(DEV).requires_grad_(True), torch.from_numpy(b).to(DEV).requires_grad_(True)) for k, b in ws]
    costs = rnnt_amd.fc_joint_loss(t_yt, t_yp, t_ws, torch.from_numpy(lab).to(DEV), torch.from_numpy(il).to(DEV),
                                   torch.from_numpy(ll).to(DEV), blank_label=blank)
    assert costs.shape == (B,)
    costs.mean().backward()   # Keras mean (keras_model.py:215-217)
    np.testing.assert_allclose(costs.detach().cpu().numpy(), co, rtol=COST_RTOL)
    assert np.abs(t_yt.grad.cpu().numpy() - dt).max() < GRAD_ATOL
    assert np.abs(t_yp.grad.cpu().numpy() - dp).max() < GRAD_ATOL
    assert not t_yt.grad[:, :2].any()
    for (k, b), (dk, db) in zip(t_ws, dws):
        assert np.abs(k.grad.cpu().numpy() - dk).max() < GRAD_ATOL
        assert np.abs(b.grad.cpu().numpy() - db).max() < GRAD_ATOL


def test_fc_joint_equals_dense_stack_on_the_tiled_tensor():
    """fc_joint against the literal torch composition on concatenate_lambda's output."""
    rng = np.random.default_rng(8)
    B, T, U1, He, Hd = 2, 7, 4, 6, 3
    yt = torch.from_numpy(rng.standard_normal((B, T, He)).astype(np.float32)).to(DEV)
    yp = torch.from_numpy(rng.standard_normal((B, U1, Hd)).astype(np.float32)).to(DEV)
    ws = [(torch.from_numpy(k).to(DEV), torch.from_numpy(b).to(DEV)) for k, b in _weights(rng, [He + Hd, 8, 8, 5])]
    x = rnnt_amd.concatenate_lambda([yt, yp])
    for i, (k, b) in enumerate(ws):
        x = x @ k + b
        if i < len(ws) - 1:
            x = torch.tanh(x)
    out = rnnt_amd.fc_joint(yt, yp, ws)
    assert out.shape == (B, T, U1, 5)
    assert (out - x).abs().max().item() < 1e-5


# M, K, N: the reference sample's last layer (100 -> 62), widths that are not multiples of anything, the limits,
# a single row, many row tiles with a ragged tail
DENSE_SHAPES = [(1000, 100, 62), (77, 7, 5), (4097, 128, 128), (1, 3, 2), (333, 33, 100), (50000, 100, 62), (260, 64, 32),
                # the weight gradient's other forms: odd widths above 64 (element-wise loads, four accumulators down),
                # two accumulators down and two across, 128 columns split over two waves with an odd K
                (500, 99, 50), (3000, 64, 40), (129, 127, 65), (2, 100, 62)]


@pytest.mark.parametrize("M,K,N", DENSE_SHAPES)
@pytest.mark.parametrize("act", ["linear", "tanh"])
def test_joint_dense_matches_float64(M, K, N, act):
    rng = np.random.default_rng(M + K + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    k = rng.uniform(-0.3, 0.3, (K, N)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, N).astype(np.float32)
    dy = rng.standard_normal((M, N)).astype(np.float32)
    z = x.astype(np.float64) @ k.astype(np.float64) + b
    y = np.tanh(z) if act == "tanh" else z
    g = dy.astype(np.float64) * ((1 - y ** 2) if act == "tanh" else 1.0)
    tx = torch.from_numpy(x).to(DEV).requires_grad_(True)
    tk = torch.from_numpy(k).to(DEV).requires_grad_(True)
    tb = torch.from_numpy(b).to(DEV).requires_grad_(True)
    out = rnnt_amd.joint_dense(tx, tk, tb, act)
    out.backward(torch.from_numpy(dy).to(DEV))
    assert np.abs(out.detach().cpu().numpy() - y).max() < 1e-5
    assert np.abs(tx.grad.cpu().numpy() - g @ k.astype(np.float64).T).max() < 1e-5 * max(1, N) ** 0.5 * 4
    scale = max(1.0, np.sqrt(M))
    assert np.abs(tk.grad.cpu().numpy() - x.astype(np.float64).T @ g).max() < 2e-5 * scale
    assert np.abs(tb.grad.cpu().numpy() - g.sum(0)).max() < 2e-5 * scale
    # bit-reproducible
    tx.grad = tk.grad = tb.grad = None
    out2 = rnnt_amd.joint_dense(tx, tk, tb, act)
    gk = torch.autograd.grad(out2, tk, torch.from_numpy(dy).to(DEV))[0]
    out3 = rnnt_amd.joint_dense(tx, tk, tb, act)
    gk2 = torch.autograd.grad(out3, tk, torch.from_numpy(dy).to(DEV))[0]
    assert torch.equal(gk, gk2) and torch.equal(out2, out3)


def test_joint_dense_wide_layers_use_the_library_and_bad_arguments_raise():
    x = torch.randn(10, 200, device=DEV)
    k, b = torch.randn(200, 30, device=DEV), torch.zeros(30, device=DEV)
    assert (rnnt_amd.joint_dense(x, k, b) - (x @ k + b)).abs().max().item() < 1e-4
    with pytest.raises(ValueError):
        rnnt_amd.joint_dense(x, torch.randn(100, 30, device=DEV), b)
    with pytest.raises(ValueError):
        rnnt_amd.joint_dense(x, k, b, "relu")
    L = _lib.lib()
    assert L.rnnt_mi355x_dense_forward(_p(x), _p(k), None, 10, 200, 30, 0, _p(x), None) == 2   # K beyond the limit
    assert L.rnnt_mi355x_dense_forward(_p(x), _p(k), None, 10, 100, 30, 2, _p(x), None) == 1   # unknown activation
