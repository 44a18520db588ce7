"""Parity of the HIP path (through the C ABI) with the float64 oracle.

Bars (north star): cost within 1e-5 relative, gradient within 1e-4 absolute.
The HIP lattice is accumulated in float64, so it is held to a tighter 2e-5
absolute on the gradient where the oracle is affordable."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import rnnt_oracle as orc
from tests.hip_runner import HipRun, make_batch

pytestmark = pytest.mark.gpu

COST_RTOL = 1e-5
GRAD_ATOL = 1e-4
GRAD_ATOL_TIGHT = 2e-5


def _check(x, lab, il, ll, blank, gs=None, atol=GRAD_ATOL_TIGHT):
    co, go = orc.rnnt_loss_and_grad(x, lab, il, ll, blank, gs)
    run = HipRun(x, lab, il, ll, blank, gs)
    assert run.loss_grad() == 0
    torch.cuda.synchronize()
    assert run.flags() == (0, 0)
    c = run.costs.cpu().numpy()
    g = run.x.cpu().numpy()
    np.testing.assert_allclose(c, co, rtol=COST_RTOL)
    err = np.abs(g - go).max()
    assert err < atol, err
    # padded cells are exact zeros
    B, T, U1, _ = x.shape
    for b in range(B):
        assert not g[b, il[b]:].any() and not g[b, :, ll[b] + 1:].any()
    # the two all-reduce scalars
    s = run.scalars()
    np.testing.assert_allclose(s[0], co.sum(), rtol=COST_RTOL)
    np.testing.assert_allclose(s[1], (go ** 2).sum(), rtol=1e-3, atol=1e-12)
    return run, co, go


def test_known_answer_vector():
    K = orc.KNOWN_ANSWER
    x = np.array(K["logits"], np.float32)
    for blank, key in ((0, "blank0"), (4, "blank4")):
        run = HipRun(x, K["labels"], [2], [2], blank)
        assert run.loss_grad() == 0
        torch.cuda.synchronize()
        assert abs(run.costs.item() - K[key + "_cost"]) < COST_RTOL * K[key + "_cost"]
        np.testing.assert_allclose(run.x.cpu().numpy(), np.array(K[key + "_grad"]), atol=2e-6)


def test_known_answer_batch_vector():
    """Published costs and gradient of the public B=2, T=4, U=2, V=3 vector."""
    x, lab, il, ll, blank, costs, grad = orc.known_answer_batch()
    run = HipRun(x.astype(np.float32), lab, il, ll, blank)
    assert run.loss_grad() == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(run.costs.cpu().numpy(), costs, rtol=COST_RTOL)
    np.testing.assert_allclose(run.x.cpu().numpy(), grad, atol=2e-6)


def test_golden_fixtures(golden_dir):
    files = [f for f in sorted(glob.glob(os.path.join(golden_dir, "*.npz"))) if not f.endswith("known_answer.npz")]
    assert len(files) >= 8
    for f in files:
        z = np.load(f)
        run = HipRun(z["logits"], z["labels"], z["in_len"], z["lab_len"], int(z["blank"]), z["grad_scale"])
        assert run.loss_grad() == 0
        torch.cuda.synchronize()
        np.testing.assert_allclose(run.costs.cpu().numpy(), z["costs"], rtol=COST_RTOL, err_msg=f)
        np.testing.assert_allclose(run.x.cpu().numpy(), z["grad"], atol=GRAD_ATOL_TIGHT, err_msg=f)


# (B, T, U, V, blank): covers every (VEC, lanes-per-row, chunks) instantiation of the row kernels
# and every columns-per-lane variant of the sweep (U+1 <= 64, 128, 256, 512, 1024).
SHAPES = [
    (3, 9, 5, 5, 0),        # VEC1 L8
    (2, 8, 4, 13, 12),      # VEC1 L16
    (2, 8, 4, 29, 3),       # VEC1 L32
    (2, 8, 4, 63, 62),      # VEC1 L64 nc1
    (2, 6, 3, 127, 5),      # VEC1 L64 nc2
    (2, 5, 3, 301, 300),    # VEC1 L64 nc4 (5 chunks per lane, masked tail)
    (2, 8, 4, 14, 13),      # VEC2 L8
    (2, 8, 4, 30, 0),       # VEC2 L16
    (4, 20, 10, 62, 39),    # VEC2 L32: the reference's 61-phone + blank=39 case (util.py:72)
    (2, 6, 3, 126, 1),      # VEC2 L64 nc1
    (2, 6, 3, 250, 7),      # VEC2 L64 nc2
    (2, 5, 3, 1022, 9),     # VEC2 L64 nc4 (8 chunks per lane)
    (3, 12, 6, 40, 39),     # VEC4 L16: 39-phone TIMIT + blank
    (2, 8, 4, 32, 31),      # VEC4 L8
    (2, 8, 4, 128, 0),      # VEC4 L32
    (2, 8, 4, 256, 255),    # VEC4 L64 nc1
    (2, 8, 4, 512, 511),    # VEC4 L64 nc2 (config 5 vocabulary)
    (2, 7, 3, 1024, 1023),  # VEC4 L64 nc4 (config 3 vocabulary)
    (2, 4, 2, 4096, 4095),  # VEC4 L64 nc4 x 4 passes (config 4 vocabulary)
    (1, 4, 2, 5000, 77),    # VEC4, chunk count not a multiple of the pass width
    # the span forms of the two passes (rows that are whole multiples of 512 bytes, up to 4096 logits): every
    # (lanes per row, chunks per lane) instantiation, row counts that are no multiple of a block's rows
    (2, 33, 9, 256, 0),     # span L16
    (2, 20, 9, 384, 3),     # span L32, 96 of 128 chunks
    (2, 50, 30, 512, 0),    # span L32 full
    (3, 45, 17, 768, 767),  # span L64 nc4, 192 of 256 chunks, ragged
    (3, 45, 17, 1024, 7),   # span L64 nc4, ragged
    (2, 12, 7, 2048, 2047), # span L64 nc8
    (2, 9, 5, 3072, 1),     # span L64 nc16, 768 of 1024 chunks
    (2, 11, 6, 4096, 17),   # span L64 nc16
    # the flat form of the gradient pass (any other 16-byte aligned row of 32..4096 logits): rows that straddle
    # waves and blocks, blocks that hold hundreds of short rows, several blocks per utterance
    (2, 33, 9, 132, 0),
    (3, 40, 20, 200, 199),
    (2, 30, 12, 300, 5),
    (2, 37, 11, 1000, 999),
    (2, 9, 5, 3000, 1),
    (1, 4, 2, 4092, 4091),
    (2, 70, 50, 36, 35),    # 144-byte rows: 1366 rows per block
    (3, 160, 90, 100, 99),  # 400-byte rows, three blocks per utterance, ragged
    (2, 30, 70, 12, 11),    # sweep K=2
    (2, 20, 150, 8, 0),     # sweep K=4
    (1, 12, 300, 4, 3),     # sweep K=8
    (1, 5, 511, 4, 0),      # U+1 = 512: eight columns per lane
    (1, 4, 700, 5, 2),      # sixteen columns per lane
    (1, 3, 1023, 3, 0),     # maximum supported U+1 = 1024
    (2, 1, 6, 10, 9),       # T = 1
    (2, 9, 0, 10, 2),       # U = 0 (no labels; labels pointer unused)
    (1, 1, 0, 3, 1),        # single cell
]


@pytest.mark.parametrize("B,T,U,V,blank", SHAPES)
def test_shapes_against_oracle(B, T, U, V, blank):
    x, lab, il, ll = make_batch(B, T, U, V, blank, seed=B * 1000 + T * 100 + U * 10 + V)
    gs = np.linspace(0.5, 1.5, B).astype(np.float32)
    _check(x, lab, il, ll, blank, gs)


@pytest.mark.parametrize("V,blank", [(1024, 1023), (62, 39), (512, 0)])
def test_very_short_utterance_next_to_a_long_one(V, blank):
    """Most frames of the short utterance lie far beyond its lattice: the first pass leaves the records of frames
    the sweeps can never reach unwritten (the workspace is poisoned with NaN here), the gradient pass zero-fills."""
    B, T, U = 3, 130, 6
    x, lab, _, _ = make_batch(B, T, U, V, blank, seed=77, ragged=False)
    il = np.array([T, 9, 40], np.int32)
    ll = np.array([U, 2, 0], np.int32)
    _check(x, lab, il, ll, blank)


def test_config1_shape():
    """BASELINE config 1: B=8 T=150 U=40 V=62, blank=39."""
    x, lab, il, ll = make_batch(8, 150, 40, 62, 39, seed=0xC0FFEE + 1, ragged=False)
    _check(x, lab, il, ll, 39)


def test_long_lattice_accuracy():
    """One config-2-sized utterance (T=500, U=100, V=62) and one config-3-sized
    one (T=400, U=100, V=1024): the regime where a float32 lattice drifts by
    ~1e-3; the float64-accumulated sweep must stay inside the bar."""
    for (T, U, V, blank, seed) in ((500, 100, 62, 39, 5), (400, 100, 1024, 1023, 6)):
        x, lab, il, ll = make_batch(2, T, U, V, blank, seed=seed, ragged=True)
        _check(x, lab, il, ll, blank)


def test_peaked_and_log_prob_inputs():
    """Large-magnitude logits (confident model) and already-log-softmaxed
    inputs (what the reference passes, util.py:66/109)."""
    x, lab, il, ll = make_batch(3, 40, 12, 64, 63, seed=9, scale=15.0)
    _check(x, lab, il, ll, 63)
    lp = orc.log_softmax(x).astype(np.float32)
    run, co, go = _check(lp, lab, il, ll, 63)
    co2, _ = orc.rnnt_loss_and_grad(x, lab, il, ll, 63)
    np.testing.assert_allclose(co, co2, rtol=1e-6)


def test_forward_then_out_of_place_backward_matches_fused():
    x, lab, il, ll = make_batch(3, 17, 9, 96, 40, seed=21)
    gs = np.array([1.0, 0.5, 2.0], np.float32)
    fused = HipRun(x, lab, il, ll, 40, gs)
    assert fused.loss_grad() == 0
    split = HipRun(x, lab, il, ll, 40, gs)
    assert split.forward() == 0
    out = torch.full_like(split.x, float("nan"))
    assert split.backward(out) == 0
    torch.cuda.synchronize()
    assert torch.equal(split.costs, fused.costs)
    assert torch.equal(out, fused.x)
    assert torch.equal(split.x, torch.from_numpy(x).to(split.x.device))  # logits untouched
    # a second backward from the same state with another scale
    split.gs = None
    assert split.backward(out) == 0
    torch.cuda.synchronize()
    _, go = orc.rnnt_loss_and_grad(x, lab, il, ll, 40)
    assert np.abs(out.cpu().numpy() - go).max() < GRAD_ATOL_TIGHT


def test_label_equal_to_blank_is_sum_of_terms():
    x, lab, il, ll = make_batch(2, 6, 4, 8, 3, seed=33, ragged=False)
    lab[0, 1] = 3
    _check(x, lab, il, ll, 3)


def test_device_side_validation_flags():
    x, lab, il, ll = make_batch(3, 6, 4, 8, 0, seed=1, ragged=False)
    bad_len = il.copy()
    bad_len[1] = 7  # > maxT
    run = HipRun(x, lab, bad_len, ll, 0)
    assert run.loss_grad() == 0
    rc, flags = run.flags()
    assert rc == 4 and flags & 1
    c = run.costs.cpu().numpy()
    g = run.x.cpu().numpy()
    assert np.isnan(c[1]) and np.isfinite(c[[0, 2]]).all()
    assert not g[1].any()  # the malformed utterance gets a zero gradient, the others are untouched by it
    co, go = orc.rnnt_loss_and_grad(x[[0, 2]], lab[[0, 2]], il[[0, 2]], ll[[0, 2]], 0)
    np.testing.assert_allclose(c[[0, 2]], co, rtol=COST_RTOL)
    assert np.abs(g[[0, 2]] - go).max() < GRAD_ATOL_TIGHT
    bad_lab = lab.copy()
    bad_lab[2, 0] = 8  # == V
    run = HipRun(x, bad_lab, il, ll, 0)
    assert run.loss_grad() == 0
    rc, flags = run.flags()
    assert rc == 4 and flags & 2
    assert np.isfinite(run.x.cpu().numpy()).all()


@pytest.mark.parametrize("U", [130, 200, 330])
def test_wide_lattices_ragged_edges_and_a_malformed_length(U):
    """More than 128 label positions: the sweeps run one wavefront per 64 columns, a band of diagonals apart.
    Utterances whose last diagonal ends inside a band, one with no labels, one with a single frame, and one with an
    impossible length (its block leaves before the first barrier) in the same batch."""
    B, T, V, blank = 6, 70, 6, 2
    x, lab, _, _ = make_batch(B, T, U, V, blank, seed=U, ragged=False)
    il = np.array([T, 33, 64, 1, T, 17], np.int32)
    ll = np.array([U, 127, 0, U, U - 1, 64], np.int32)
    _check(x, lab, il, ll, blank)
    bad = il.copy()
    bad[1] = T + 1
    run = HipRun(x, lab, bad, ll, blank)
    assert run.loss_grad() == 0
    rc, flags = run.flags()
    assert rc == 4 and flags & 1
    c = run.costs.cpu().numpy()
    keep = [0, 2, 3, 4, 5]
    assert np.isnan(c[1]) and not run.x.cpu().numpy()[1].any()
    co, go = orc.rnnt_loss_and_grad(x[keep], lab[keep], il[keep], ll[keep], blank)
    np.testing.assert_allclose(c[keep], co, rtol=COST_RTOL)
    assert np.abs(run.x.cpu().numpy()[keep] - go).max() < GRAD_ATOL_TIGHT


def test_deterministic_and_batch_independent():
    """Same inputs twice -> bit-identical outputs; an utterance's result does
    not depend on what else is in the batch."""
    x, lab, il, ll = make_batch(4, 25, 11, 62, 39, seed=77)
    a = HipRun(x, lab, il, ll, 39)
    b = HipRun(x, lab, il, ll, 39)
    assert a.loss_grad() == 0 and b.loss_grad() == 0
    torch.cuda.synchronize()
    assert torch.equal(a.costs, b.costs) and torch.equal(a.x, b.x)
    T2, U2 = int(il[2]), int(ll[2])
    solo = HipRun(x[2:3, :T2, :U2 + 1].copy(), lab[2:3, :U2], [T2], [U2], 39)
    assert solo.loss_grad() == 0
    torch.cuda.synchronize()
    assert torch.equal(solo.costs[0], a.costs[2])
    assert torch.equal(solo.x[0], a.x[2, :T2, :U2 + 1])


def test_unaligned_base_pointer_falls_back_to_narrower_loads():
    x, lab, il, ll = make_batch(2, 6, 3, 64, 63, seed=4)
    co, go = orc.rnnt_loss_and_grad(x, lab, il, ll, 63)
    run = HipRun(x, lab, il, ll, 63)
    big = torch.empty(run.x.numel() + 4, dtype=torch.float32, device=run.dev)
    for off in (1, 2):  # 4-byte and 8-byte aligned only
        view = big[off:off + run.x.numel()].view_as(run.x)
        view.copy_(torch.from_numpy(x))
        run.x = view
        assert run.loss_grad() == 0
        torch.cuda.synchronize()
        assert np.abs(view.cpu().numpy() - go).max() < GRAD_ATOL_TIGHT


@pytest.mark.parametrize("V,blank", [(1024, 1023), (62, 39), (40, 39), (100, 7), (300, 299)])
def test_batch_strided_entry_points(V, blank):
    """rnnt_mi355x_forward_strided / backward_strided on a crop along T of a larger
    tensor (utterances T0*U1*V elements apart), in place and out of place."""
    import ctypes
    from rnnt_amd import _lib
    B, T0, U, shift = 3, 11, 4, 2
    T = T0 - shift
    rng = np.random.default_rng(V)
    full = rng.standard_normal((B, T0, U + 1, V)).astype(np.float32)
    lab = rng.integers(0, min(V - 1, 39), (B, U)).astype(np.int32)
    il = np.array([T, T - 3, 1], np.int32)
    ll = np.array([U, 2, 0], np.int32)
    gs = np.array([1.0, 0.5, 2.0], np.float32)
    co, go = orc.rnnt_loss_and_grad(full[:, shift:], lab, il, ll, blank, gs)
    dev = torch.device("cuda:0")
    t_full = torch.from_numpy(full).to(dev)
    view = t_full[:, shift:]
    run = HipRun(full[:, shift:], lab, il, ll, blank, gs)          # only for its helper buffers (labels, lengths, ws)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    L = _lib.lib()
    stride = T0 * (U + 1) * V
    rc = L.rnnt_mi355x_forward_strided(p(view), stride, p(run.lab), p(run.il), p(run.ll), B, T, U + 1, V, blank,
                                       p(run.costs), p(run.ws), run.stream)
    assert rc == 0
    out = torch.full((B, T, U + 1, V), float("nan"), device=dev)
    rc = L.rnnt_mi355x_backward_strided(p(view), stride, p(out), T * (U + 1) * V, p(run.lab), p(run.il), p(run.ll),
                                        B, T, U + 1, V, blank, p(run.gs), p(run.ws), run.stream, run.prune)
    assert rc == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(run.costs.cpu().numpy(), co, rtol=COST_RTOL)
    assert np.abs(out.cpu().numpy() - go).max() < GRAD_ATOL_TIGHT
    assert torch.equal(t_full.cpu(), torch.from_numpy(full))       # logits untouched so far
    rc = L.rnnt_mi355x_backward_strided(p(view), stride, p(view), stride, p(run.lab), p(run.il), p(run.ll),
                                        B, T, U + 1, V, blank, p(run.gs), p(run.ws), run.stream, run.prune)
    assert rc == 0
    torch.cuda.synchronize()
    got = t_full.cpu().numpy()
    assert np.abs(got[:, shift:] - go).max() < GRAD_ATOL_TIGHT
    assert np.array_equal(got[:, :shift], full[:, :shift])        # frames outside the crop are never written
    # a stride smaller than one utterance is rejected
    assert L.rnnt_mi355x_forward_strided(p(view), T * (U + 1) * V - 1, p(run.lab), p(run.il), p(run.ll), B, T, U + 1,
                                         V, blank, p(run.costs), p(run.ws), run.stream) == _lib.INVALID_VALUE


def test_random_shapes_fuzz():
    """40 seeded random problems (shape, vocabulary, blank position, ragged lengths, logit scale)
    through rnnt_mi355x_loss_grad against the float64 oracle."""
    rng = np.random.default_rng(20261111)
    for case in range(40):
        B = int(rng.integers(1, 6))
        T = int(rng.integers(1, 40))
        U = int(rng.integers(0, 30)) if case % 7 else int(rng.integers(60, 140))
        V = int(rng.choice([2, 3, 7, 16, 31, 40, 62, 64, 65, 100, 128, 257, 300, 512, 777, 1024, 1500]))
        blank = int(rng.integers(0, V))
        scale = float(rng.choice([0.3, 1.0, 4.0, 12.0]))
        x, lab, il, ll = make_batch(B, T, U, V, blank, seed=1000 + case, ragged=bool(case % 3), scale=scale)
        gs = rng.uniform(0.2, 2.0, B).astype(np.float32) if case % 2 else None
        try:
            _check(x, lab, il, ll, blank, gs)
        except AssertionError as e:
            raise AssertionError(f"case {case}: B={B} T={T} U={U} V={V} blank={blank} scale={scale}: {e}")


@pytest.mark.parametrize("B,T,U,V,blank", [(2, 9, 4, 40, 39), (2, 7, 3, 300, 0)])
def test_masked_vocabulary_entries(B, T, U, V, blank):
    """Logits of -inf (masked symbols that are neither the blank nor a needed label): probability exactly 0,
    gradient exactly 0 there, everything else unaffected."""
    x, lab, il, ll = make_batch(B, T, U, V, blank, seed=77)
    rng = np.random.default_rng(5)
    mask = rng.random(x.shape) < 0.15
    mask[..., blank] = False
    for b in range(B):
        mask[b][..., lab[b]] = False
    x = x.copy()
    x[mask] = -np.inf
    with np.errstate(invalid="ignore"):
        co, go = orc.rnnt_loss_and_grad(x, lab, il, ll, blank)
    assert np.isfinite(co).all() and np.isfinite(go).all()
    run = HipRun(x, lab, il, ll, blank)
    assert run.loss_grad() == 0
    torch.cuda.synchronize()
    g = run.x.cpu().numpy()
    np.testing.assert_allclose(run.costs.cpu().numpy(), co, rtol=COST_RTOL)
    assert np.isfinite(g).all() and np.abs(g - go).max() < GRAD_ATOL_TIGHT
    assert not g[mask].any()
