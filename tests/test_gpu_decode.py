"""Decoder inner step (joint + Util.softmax(is_log=True), reference
rnnt_decoder.py:156-158,200-202) batched on the MI355X, against the float64
restatement of Util.softmax.  Tolerance 1e-5 absolute on log-probabilities."""
import numpy as np
import pytest
import torch

import rnnt_amd
from oracle import rnnt_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("V", [1, 5, 62, 64, 65, 1024, 2048, 2049, 5000])
def test_joint_log_softmax_rows(V):
    rng = np.random.default_rng(V)
    enc = (rng.standard_normal((7, V)) * 4).astype(np.float32)
    pred = (rng.standard_normal((7, V)) * 4).astype(np.float32)
    out = rnnt_amd.joint_log_softmax(torch.from_numpy(enc).to(DEV), torch.from_numpy(pred).to(DEV)).cpu().numpy()
    for n in range(7):
        np.testing.assert_allclose(out[n], orc.decoder_step_logp(enc[n], pred[n]), atol=1e-5)
    alone = rnnt_amd.joint_log_softmax(torch.from_numpy(enc).to(DEV)).cpu().numpy()
    for n in range(7):
        np.testing.assert_allclose(alone[n], orc.util_softmax(enc[n], is_log=True), atol=1e-5)


def test_beam_of_hypotheses_shares_encoder_rows():
    """A beam: 3 encoder frames, 5 hypotheses' prediction vectors, 11 (frame, hypothesis) pairs in one launch."""
    rng = np.random.default_rng(1)
    V = 62
    enc = rng.standard_normal((3, V)).astype(np.float32)
    pred = rng.standard_normal((5, V)).astype(np.float32)
    fi = np.array([0, 0, 0, 1, 1, 2, 2, 2, 2, 0, 1])
    gi = np.array([0, 1, 2, 0, 4, 3, 3, 1, 0, 4, 2])
    out = rnnt_amd.joint_log_softmax(torch.from_numpy(enc).to(DEV), torch.from_numpy(pred).to(DEV),
                                     torch.from_numpy(fi).to(DEV), torch.from_numpy(gi).to(DEV)).cpu().numpy()
    assert out.shape == (11, V)
    for n in range(11):
        np.testing.assert_allclose(out[n], orc.decoder_step_logp(enc[fi[n]], pred[gi[n]]), atol=1e-5)
    # only one side indexed
    out2 = rnnt_amd.joint_log_softmax(torch.from_numpy(enc).to(DEV), torch.from_numpy(pred[:3]).to(DEV),
                                      enc_index=torch.tensor([2, 1, 0])).cpu().numpy()
    for n in range(3):
        np.testing.assert_allclose(out2[n], orc.decoder_step_logp(enc[2 - n], pred[n]), atol=1e-5)


def test_bad_arguments():
    enc, pred = torch.zeros(3, 8, device=DEV), torch.zeros(2, 8, device=DEV)
    with pytest.raises(IndexError):
        rnnt_amd.joint_log_softmax(enc, pred, torch.tensor([0, 3]), torch.tensor([0, 1]))
    with pytest.raises(IndexError):
        rnnt_amd.joint_log_softmax(enc, pred, torch.tensor([0, 1]), torch.tensor([0, -1]))
    with pytest.raises(IndexError):                                # indices already on the device: checked by the kernel
        rnnt_amd.joint_log_softmax(enc, pred, torch.tensor([0, 3], device=DEV), torch.tensor([0, 1], device=DEV))
    ok = rnnt_amd.joint_log_softmax(enc, pred, torch.tensor([0, 2], device=DEV), torch.tensor([1, 1], device=DEV))
    assert ok.shape == (2, 8)
    with pytest.raises(ValueError):
        rnnt_amd.joint_log_softmax(enc, pred)                      # row counts differ, no indices
    with pytest.raises(ValueError):
        rnnt_amd.joint_log_softmax(enc, torch.zeros(3, 9, device=DEV))
    with pytest.raises(ValueError):
        rnnt_amd.joint_log_softmax(torch.zeros(3, 8))              # CPU tensor
    assert rnnt_amd.joint_log_softmax(enc, pred, torch.zeros(0, dtype=torch.int32), torch.zeros(0, dtype=torch.int32)).shape == (0, 8)


def test_softmax_mirrors_util_softmax():
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(62) * 3).astype(np.float32)
    tx = torch.from_numpy(x).to(DEV)
    np.testing.assert_allclose(rnnt_amd.softmax(tx).cpu().numpy(), orc.util_softmax(x), atol=1e-6)
    np.testing.assert_allclose(rnnt_amd.softmax(tx, is_log=True).cpu().numpy(), orc.util_softmax(x, is_log=True), atol=1e-5)
    # axis=0: the reference's element-wise recursion (a vector becomes ones, is_log is dropped)
    assert rnnt_amd.softmax(tx, is_log=True, axis=0).cpu().tolist() == orc.util_softmax(x, is_log=True, axis=0)
    m = rng.standard_normal((4, 9)).astype(np.float32)
    got = rnnt_amd.softmax(torch.from_numpy(m).to(DEV), is_log=True, axis=0).cpu().numpy()
    np.testing.assert_allclose(got, np.stack(orc.util_softmax(m, is_log=True, axis=0)), atol=1e-6)
    # a matrix with axis=None is normalised over all of its elements (util.py:303-307)
    np.testing.assert_allclose(rnnt_amd.softmax(torch.from_numpy(m).to(DEV)).cpu().numpy(), orc.util_softmax(m), atol=1e-6)
    with pytest.raises(AssertionError):
        rnnt_amd.softmax(tx, axis=1)
