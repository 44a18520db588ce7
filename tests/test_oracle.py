"""Pins the float64 oracle (oracle/rnnt_oracle.py) before anything is compared
with it: public known-answer vector, brute-force path enumeration, finite
differences, diagonal-vs-plain recursion, and the committed golden fixtures."""
import glob
import os

import numpy as np
import pytest

from oracle import rnnt_oracle as orc


def _ka():
    K = orc.KNOWN_ANSWER
    return np.array(K["logits"]), np.array(K["labels"]), K


@pytest.mark.parametrize("blank,key", [(0, "blank0"), (4, "blank4")])
def test_known_answer_vector(blank, key):
    x, lab, K = _ka()
    costs, grad = orc.rnnt_loss_and_grad(x, lab, [2], [2], blank)
    assert abs(costs[0] - K[key + "_cost"]) < 1e-8
    np.testing.assert_allclose(grad, np.array(K[key + "_grad"]), atol=2e-8)


@pytest.mark.parametrize("blank", [0, 4])
def test_known_answer_rederived_independently(blank):
    """The public vector is not taken on trust: brute-force path enumeration
    and finite differences must give the same numbers."""
    x, lab, K = _ka()
    key = "blank0" if blank == 0 else "blank4"
    assert abs(orc.brute_force_cost(x[0], lab[0], 2, 2, blank) - K[key + "_cost"]) < 1e-8
    ng = orc.numeric_grad(x, lab, [2], [2], blank)
    np.testing.assert_allclose(ng, np.array(K[key + "_grad"]), atol=5e-8)


def test_known_answer_batch_vector():
    """Second public vector (B=2, T=4, U=2, V=3): published costs and gradient,
    six decimals, computed upstream in float32."""
    x, lab, il, ll, blank, costs, grad = orc.known_answer_batch()
    c, g = orc.rnnt_loss_and_grad(x, lab, il, ll, blank)
    np.testing.assert_allclose(c, costs, rtol=2e-7)
    np.testing.assert_allclose(g, grad, atol=1e-6)
    for b in range(2):  # and re-derived by enumerating every alignment
        assert abs(orc.brute_force_cost(x[b], lab[b], 4, 2, blank) - costs[b]) < 1e-6


@pytest.mark.parametrize("T,U,V,blank", [(1, 0, 3, 0), (1, 3, 4, 2), (4, 0, 3, 1), (3, 3, 5, 4), (4, 2, 6, 0), (2, 4, 3, 1)])
def test_brute_force_paths(T, U, V, blank):
    rng = np.random.default_rng(T * 100 + U * 10 + V)
    x = rng.standard_normal((1, T, U + 1, V)) * 2
    lab = rng.integers(0, V - 1, size=(1, max(U, 1)))
    lab[lab >= blank] += 1
    lab = lab[:, :U]
    costs, _ = orc.rnnt_loss_and_grad(x, lab.reshape(1, U), [T], [U], blank)
    assert abs(costs[0] - orc.brute_force_cost(x[0], lab[0], T, U, blank)) < 1e-10


def test_finite_differences_ragged():
    rng = np.random.default_rng(7)
    B, T, U, V, blank = 2, 4, 3, 5, 2
    x = rng.standard_normal((B, T, U + 1, V))
    lab = np.array([[0, 1, 3], [4, 4, 0]])
    il, ll = [4, 3], [3, 1]
    _, g = orc.rnnt_loss_and_grad(x, lab, il, ll, blank)
    ng = orc.numeric_grad(x, lab, il, ll, blank)
    np.testing.assert_allclose(g, ng, atol=1e-7)
    assert np.all(g[1, 3:] == 0) and np.all(g[1, :, 2:] == 0)  # padded cells


def test_diag_recursion_equals_plain():
    rng = np.random.default_rng(3)
    lpb = -rng.uniform(0.1, 3, (9, 6))
    lpl = -rng.uniform(0.1, 3, (9, 5))
    a0, b0, l0 = orc.alpha_beta(lpb, lpl)
    a1, b1, l1 = orc.alpha_beta_diag(lpb, lpl)
    np.testing.assert_allclose(a0, a1, atol=1e-12)
    np.testing.assert_allclose(b0, b1, atol=1e-12)
    assert abs(l0 - l1) < 1e-12 and abs(b0[0, 0] - l0) < 1e-10


def test_log_softmax_idempotent_inputs():
    """Passing log-probs (what the reference does, util.py:66/109) or raw
    logits gives the same costs and the same gradient."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, 5, 4, 7))
    lab = rng.integers(1, 7, (2, 3))
    c0, g0 = orc.rnnt_loss_and_grad(x, lab, [5, 4], [3, 2], 0)
    c1, g1 = orc.rnnt_loss_and_grad(orc.log_softmax(x), lab, [5, 4], [3, 2], 0)
    np.testing.assert_allclose(c0, c1, atol=1e-12)
    np.testing.assert_allclose(g0, g1, atol=1e-12)


def test_gradient_rows_sum_to_zero_and_blank_occupancy():
    rng = np.random.default_rng(5)
    T, U, V = 6, 4, 9
    x = rng.standard_normal((1, T, U + 1, V))
    lab = rng.integers(1, V, (1, U))
    _, g = orc.rnnt_loss_and_grad(x, lab, [T], [U], 0)
    np.testing.assert_allclose(g.sum(-1), 0, atol=1e-12)


def test_label_equal_to_blank_sums_both_terms():
    x = np.array(orc.KNOWN_ANSWER["logits"])
    lab = np.array([[0, 2]])  # first label equals blank index 0
    c, g = orc.rnnt_loss_and_grad(x, lab, [2], [2], 0)
    assert np.isfinite(c).all() and abs(g.sum()) < 1e-12


def test_lambda_funcs_follow_reference_contract():
    rng = np.random.default_rng(2)
    B, T0, U, V = 2, 7, 3, 6
    yt = rng.standard_normal((B, T0, V))
    yp = rng.standard_normal((B, U + 1, V))
    lab = rng.integers(0, V - 1, (B, U))
    il = np.array([[7], [6]])
    ll = np.array([[3], [2]])
    costs, dt, dp = orc.rnnt_lambda_func(yt, yp, lab, il, ll, blank_label=V - 1)
    joint = yt[:, :, None, :] + yp[:, None, :, :]
    c2, da = orc.rnnt_lambda_func_v2(joint, lab, il, ll, blank_label=V - 1)
    np.testing.assert_allclose(costs, c2, atol=1e-12)
    assert np.all(dt[:, :2] == 0) and np.all(da[:, :2] == 0)  # cropped frames get no gradient
    np.testing.assert_allclose(dt, da.sum(2), atol=1e-12)
    np.testing.assert_allclose(dp, da.sum(1), atol=1e-12)
    # direct call on the cropped tensor with lengths-2
    c3, _ = orc.rnnt_loss_and_grad(joint[:, 2:], lab, il.reshape(-1) - 2, ll.reshape(-1), V - 1)
    np.testing.assert_allclose(costs, c3, atol=1e-12)


def test_golden_fixtures_match_oracle(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "*.npz")))
    assert len(files) >= 9
    for f in files:
        z = np.load(f)
        if os.path.basename(f) == "known_answer.npz":
            for blank, key in ((0, "blank0"), (4, "blank4")):
                c, g = orc.rnnt_loss_and_grad(z["logits"], z["labels"], z["in_len"], z["lab_len"], blank)
                assert abs(c[0] - float(z[key + "_cost"])) < 1e-6
                np.testing.assert_allclose(g, z[key + "_grad"], atol=1e-7)
            continue
        c, g = orc.rnnt_loss_and_grad(z["logits"], z["labels"], z["in_len"], z["lab_len"], int(z["blank"]), z["grad_scale"])
        np.testing.assert_allclose(c, z["costs"], rtol=1e-12)
        np.testing.assert_allclose(g, z["grad"], atol=1e-12)


def test_fc_joint_head_gradients_by_finite_differences():
    """The FC-joint restatement (concat -> Dense/tanh stack -> lambda v2) against
    central differences of its own cost, for every input and weight."""
    rng = np.random.default_rng(77)
    B, T0, U, He, Hd, H, V = 2, 6, 2, 3, 2, 4, 5
    yt, yp = rng.standard_normal((B, T0, He)), rng.standard_normal((B, U + 1, Hd))
    ws = [(rng.standard_normal((He + Hd, H)) * 0.5, rng.standard_normal(H) * 0.1),
          (rng.standard_normal((H, V)) * 0.5, rng.standard_normal(V) * 0.1)]
    lab = rng.integers(0, V - 1, (B, U))
    il, ll = np.array([[6], [5]]), np.array([[2], [1]])
    costs, dt, dp, dws = orc.fc_joint_loss_and_grads(yt, yp, ws, lab, il, ll, blank_label=4)

    def total():
        return orc.fc_joint_loss_and_grads(yt, yp, ws, lab, il, ll, blank_label=4)[0].sum()

    def fd(arr, idx, eps=1e-6):
        old = arr[idx]
        arr[idx] = old + eps
        cp = total()
        arr[idx] = old - eps
        cm = total()
        arr[idx] = old
        return (cp - cm) / (2 * eps)

    for arr, g in ((yt, dt), (yp, dp), (ws[0][0], dws[0][0]), (ws[0][1], dws[0][1]), (ws[1][0], dws[1][0]), (ws[1][1], dws[1][1])):
        for _ in range(6):
            idx = tuple(rng.integers(0, n) for n in arr.shape)
            assert abs(fd(arr, idx) - g[idx]) < 1e-6
    assert not dt[:, :2].any()  # the two cropped frames (util.py:63)
    # a single linear layer is the additive joint of rnnt_lambda_func
    k, b = rng.standard_normal((He + Hd, V)), rng.standard_normal(V)
    c1 = orc.fc_joint_loss_and_grads(yt, yp, [(k, b)], lab, il, ll, blank_label=4)[0]
    c2 = orc.rnnt_lambda_func(yt @ k[:He], yp @ k[He:] + b, lab, il, ll, blank_label=4)[0]
    np.testing.assert_allclose(c1, c2, rtol=1e-12)


def test_util_softmax_restatement():
    rng = np.random.default_rng(9)
    x = rng.standard_normal(62) * 3
    p = orc.util_softmax(x)
    assert abs(p.sum() - 1) < 1e-12 and np.all(p > 0)
    np.testing.assert_allclose(orc.util_softmax(x, is_log=True), np.log(p), atol=1e-12)
    np.testing.assert_allclose(orc.util_softmax(x, is_log=True), orc.log_softmax(x[None])[0], atol=1e-12)
    # the axis=0 branch drops is_log and works element by element (util.py:297-301)
    assert orc.util_softmax(x, is_log=True, axis=0) == [1.0] * 62
    m = rng.standard_normal((3, 5))
    rows = orc.util_softmax(m, is_log=True, axis=0)
    for r, row in zip(rows, m):
        np.testing.assert_allclose(r, np.exp(row) / np.exp(row).sum(), atol=1e-12)
    np.testing.assert_allclose(orc.decoder_step_logp(x, x[::-1]), orc.log_softmax((x + x[::-1])[None])[0], atol=1e-12)


def test_op_fixtures_are_reproduced(golden_dir):
    """tests/golden/ops/*.npz: the lambdas, the FC-joint loss head, the decoder step, the second public vector."""
    ops = os.path.join(golden_dir, "ops")
    z = np.load(os.path.join(ops, "lambda_additive_joint.npz"))
    c, dt, dp = orc.rnnt_lambda_func(z["y_trans"], z["y_pred"], z["labels"], z["input_length"], z["label_length"],
                                     blank_label=39, grad_scale=z["grad_scale"])
    np.testing.assert_allclose(c, z["costs"], rtol=1e-12)
    np.testing.assert_allclose(dt, z["d_trans"], atol=1e-12)
    np.testing.assert_allclose(dp, z["d_pred"], atol=1e-12)
    z = np.load(os.path.join(ops, "fc_joint_two_layers.npz"))
    c, dt, dp, dws = orc.fc_joint_loss_and_grads(z["y_trans"], z["y_pred"], [(z["k1"], z["b1"]), (z["k2"], z["b2"])],
                                                 z["labels"], z["input_length"], z["label_length"], blank_label=39,
                                                 grad_scale=z["grad_scale"])
    np.testing.assert_allclose(c, z["costs"], rtol=1e-12)
    np.testing.assert_allclose(dws[1][0], z["dk2"], atol=1e-12)
    z = np.load(os.path.join(ops, "decoder_step.npz"))
    for n, (i, j) in enumerate(zip(z["enc_index"], z["pred_index"])):
        np.testing.assert_allclose(orc.decoder_step_logp(z["enc"][i], z["pred"][j]), z["logp"][n], atol=1e-12)
    z = np.load(os.path.join(ops, "known_answer_batch.npz"))
    c, g = orc.rnnt_loss_and_grad(z["logits"], z["labels"], z["in_len"], z["lab_len"], int(z["blank"]))
    np.testing.assert_allclose(c, z["published_costs"], rtol=2e-7)
    np.testing.assert_allclose(g, z["published_grad"], atol=1e-6)
