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


def _run_sanitized(tmp_path, tag, x, lab, il, ll, blank, scale=None, want_grad=True, inplace=False):
    """One case through the address + undefined-behaviour sanitizer build (oracle/Makefile `asan`), in a child
    process; returns (status, costs, grad)."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(cpu_port.__file__)), "_asan", "rnnt_cpu_asan")
    x = np.ascontiguousarray(x, np.float32)
    B, T, U1, V = x.shape
    lab = np.ascontiguousarray(np.asarray(lab, np.int32).reshape(B, -1))
    assert lab.shape[1] == max(U1 - 1, 0)
    inp, out = tmp_path / f"{tag}.in", tmp_path / f"{tag}.out"
    with open(inp, "wb") as f:
        f.write(np.array([B, T, U1, V, blank, scale is not None, want_grad, inplace], np.int32).tobytes())
        f.write(x.tobytes()); f.write(lab.tobytes())
        f.write(np.asarray(il, np.int32).tobytes()); f.write(np.asarray(ll, np.int32).tobytes())
        if scale is not None:
            f.write(np.asarray(scale, np.float32).tobytes())
    r = subprocess.run([exe, str(inp), str(out)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", OMP_NUM_THREADS="2"))
    assert r.returncode == 0 and "ERROR" not in r.stderr and "runtime error" not in r.stderr, f"{tag}: {r.stderr[-2000:]}"
    raw = open(out, "rb").read()
    status = int(np.frombuffer(raw[:4], np.int32)[0])
    costs = np.frombuffer(raw[4:4 + 4 * B], np.float32)
    grad = np.frombuffer(raw[4 + 4 * B:], np.float32).reshape(x.shape) if want_grad and status == 0 else None
    return status, costs, grad


def test_sanitizer_build_of_the_c_restatement(golden_dir, tmp_path):
    """`make -C oracle asan` (-fsanitize=address,undefined) and the golden fixtures plus the edge cases (no labels,
    one frame, ragged lengths, in place, costs only, rejected input) through it: no report, same numbers as the
    optimised build."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or not os.path.exists(subprocess.run(
            ["gcc", "-print-file-name=libasan.a"], capture_output=True, text=True).stdout.strip()):
        pytest.skip("no sanitizer runtime for gcc on this machine")
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(os.path.abspath(cpu_port.__file__)), "asan"])
    for f in sorted(glob.glob(os.path.join(golden_dir, "*.npz"))):
        if f.endswith("known_answer.npz"):
            continue
        z = np.load(f)
        st, c, g = _run_sanitized(tmp_path, os.path.basename(f)[:-4], z["logits"], z["labels"], z["in_len"], z["lab_len"],
                                  int(z["blank"]), z["grad_scale"])
        assert st == 0
        c0, g0 = cpu_port.loss_grad(z["logits"], z["labels"], z["in_len"], z["lab_len"], int(z["blank"]), z["grad_scale"])
        np.testing.assert_allclose(c, c0, rtol=2e-6, err_msg=f)
        np.testing.assert_allclose(g, g0, atol=2e-6, err_msg=f)
    rng = np.random.default_rng(17)
    # U = 0 everywhere (U1 = 1: no label array at all), T = 1, and a ragged batch that includes both
    x = rng.standard_normal((3, 5, 1, 7)).astype(np.float32)
    st, c, g = _run_sanitized(tmp_path, "u0", x, np.zeros((3, 0), np.int32), [5, 1, 3], [0, 0, 0], 2)
    co, go = orc.rnnt_loss_and_grad(x, np.zeros((3, 0), np.int32), [5, 1, 3], [0, 0, 0], 2)
    assert st == 0
    np.testing.assert_allclose(c, co, rtol=COST_RTOL); np.testing.assert_allclose(g, go, atol=GRAD_ATOL)
    x = rng.standard_normal((2, 1, 4, 6)).astype(np.float32)
    lab = rng.integers(0, 5, (2, 3)).astype(np.int32)
    st, c, g = _run_sanitized(tmp_path, "t1", x, lab, [1, 1], [3, 0], 5)
    co, go = orc.rnnt_loss_and_grad(x, lab, [1, 1], [3, 0], 5)
    assert st == 0
    np.testing.assert_allclose(c, co, rtol=COST_RTOL); np.testing.assert_allclose(g, go, atol=GRAD_ATOL)
    x = rng.standard_normal((4, 9, 6, 11)).astype(np.float32)
    lab = rng.integers(0, 10, (4, 5)).astype(np.int32); lab[lab >= 3] += 1
    il, ll = [9, 1, 4, 7], [5, 0, 5, 2]
    co, go = orc.rnnt_loss_and_grad(x, lab, il, ll, 3, [0.5, 1.0, 2.0, 0.25])
    for tag, kw in (("ragged", {}), ("inplace", {"inplace": True}), ("costs_only", {"want_grad": False})):
        st, c, g = _run_sanitized(tmp_path, tag, x, lab, il, ll, 3, [0.5, 1.0, 2.0, 0.25], **kw)
        assert st == 0
        np.testing.assert_allclose(c, co, rtol=COST_RTOL)
        if g is not None:
            np.testing.assert_allclose(g, go, atol=GRAD_ATOL)
    # rejected: a length beyond the tensor, a label outside the vocabulary
    assert _run_sanitized(tmp_path, "bad_len", x, lab, [10, 1, 4, 7], ll, 3)[0] == 2
    bad = lab.copy(); bad[0, 0] = 11
    assert _run_sanitized(tmp_path, "bad_label", x, bad, il, ll, 3)[0] == 2
