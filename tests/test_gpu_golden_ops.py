"""The HIP paths of the callers either side of the loss against the committed fixtures tests/golden/ops/*.npz
(generator: tests/golden/make_golden.py).  Bars: cost 1e-5 relative, gradients 1e-4 absolute, log-probs 1e-5."""
import os

import numpy as np
import pytest
import torch

import rnnt_amd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t.requires_grad_(True) if grad else t


@pytest.mark.parametrize("fused", [True, False])
def test_lambda_additive_joint_fixture(golden_dir, fused):
    z = np.load(os.path.join(golden_dir, "ops", "lambda_additive_joint.npz"))
    yt, yp = _t(z["y_trans"], True), _t(z["y_pred"], True)
    costs = rnnt_amd.rnnt_lambda_func([yt, yp, _t(z["labels"]), _t(z["input_length"]), _t(z["label_length"])], fused=fused)
    (costs * _t(z["grad_scale"].astype(np.float32))).sum().backward()
    np.testing.assert_allclose(costs.detach().cpu().numpy(), z["costs"], rtol=1e-5)
    assert np.abs(yt.grad.cpu().numpy() - z["d_trans"]).max() < 1e-4
    assert np.abs(yp.grad.cpu().numpy() - z["d_pred"]).max() < 1e-4


def test_fc_joint_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "ops", "fc_joint_two_layers.npz"))
    yt, yp = _t(z["y_trans"], True), _t(z["y_pred"], True)
    ws = [(_t(z["k1"], True), _t(z["b1"], True)), (_t(z["k2"], True), _t(z["b2"], True))]
    costs = rnnt_amd.fc_joint_loss(yt, yp, ws, _t(z["labels"]), _t(z["input_length"]), _t(z["label_length"]))
    (costs * _t(z["grad_scale"].astype(np.float32))).sum().backward()
    np.testing.assert_allclose(costs.detach().cpu().numpy(), z["costs"], rtol=1e-5)
    for t, key in ((yt, "d_trans"), (yp, "d_pred"), (ws[0][0], "dk1"), (ws[0][1], "db1"), (ws[1][0], "dk2"), (ws[1][1], "db2")):
        assert np.abs(t.grad.cpu().numpy() - z[key]).max() < 1e-4, key


def test_decoder_step_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "ops", "decoder_step.npz"))
    out = rnnt_amd.joint_log_softmax(_t(z["enc"]), _t(z["pred"]), _t(z["enc_index"]), _t(z["pred_index"]))
    np.testing.assert_allclose(out.cpu().numpy(), z["logp"], atol=1e-5)


def test_published_batch_vector_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "ops", "known_answer_batch.npz"))
    acts = _t(z["logits"], True)
    costs = rnnt_amd.rnnt_loss(acts, z["labels"], z["in_len"], z["lab_len"], int(z["blank"]))
    costs.sum().backward()
    np.testing.assert_allclose(costs.detach().cpu().numpy(), z["published_costs"], rtol=1e-5)
    assert np.abs(acts.grad.cpu().numpy() - z["published_grad"]).max() < 2e-6
