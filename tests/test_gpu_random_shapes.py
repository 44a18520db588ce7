"""Seeded random shapes through the three loss surfaces (materialised logits through the C ABI, crop along T inside
the library, additive joint fused into the loss), each against the float64 oracle.  The parametrised tests elsewhere
pick one shape per kernel instantiation; these draw extents, lengths, blank position, logit spread and upstream
scales together, so that combinations nobody listed (odd vocabularies with ragged tiles, a blank in the middle, label
counts next to a sweep width boundary, ...) are exercised as well.  Same bars: cost 1e-5 relative, gradient 1e-4
absolute."""
import numpy as np
import pytest
import torch

import rnnt_amd
from oracle import rnnt_oracle as orc
from tests.hip_runner import HipRun

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
COST_RTOL = 1e-5
GRAD_ATOL = 1e-4
# A one-cell utterance whose only path is near-certain costs ~1e-3 or less.  Its float32 sum of exponentials is
# 1 + tiny, good to one ulp of 1 (6e-8) whatever the implementation, and the logits of such a row are large, so the
# cost carries a few 1e-7 absolute: the relative bar needs a floor there.
COST_ATOL = 1e-6
SUM_RTOL = 6e-5  # d f / d g of the fused joint: sums of up to hundreds of joint-gradient elements (the bar is 1e-4)
SCALES = (0.3, 1.0, 3.0, 6.0)  # standard deviations the logits are drawn with
MAX_T, MAX_U1, SIZE_FACTOR = 150, 330, 1  # the soak script raises these


def _draw(seed, max_cells=40_000, max_elems=6_000_000, vmax=1200, scales=None):
    """One random problem.  Extents are drawn log-uniformly so that small and large ones are equally likely."""
    rng = np.random.default_rng(seed)

    def logint(lo, hi):
        return int(round(np.exp(rng.uniform(np.log(lo), np.log(hi + 0.49)))))

    while True:
        B = int(rng.integers(1, 6))
        T = logint(1, MAX_T)
        U = logint(1, MAX_U1) - 1
        V = logint(2, vmax)
        if B * T * (U + 1) <= max_cells * SIZE_FACTOR and B * T * (U + 1) * V <= max_elems * SIZE_FACTOR:
            break
    blank = int(rng.integers(0, V))
    lab = rng.integers(0, V - 1, (B, U)).astype(np.int32)
    lab[lab >= blank] += 1
    if rng.random() < 0.3 and U > 1:  # long runs of one symbol
        lab[:, : U // 2] = lab[:, :1]
    il = rng.integers(1, T + 1, B).astype(np.int32)
    ll = rng.integers(0, U + 1, B).astype(np.int32)
    k = int(rng.integers(0, B))
    il[k], ll[k] = T, U
    scale = float(rng.choice(scales or SCALES))
    gs = rng.uniform(0.25, 2.0, B).astype(np.float32)
    return rng, B, T, U, V, blank, lab, il, ll, scale, gs


@pytest.mark.parametrize("seed", range(1000, 1060))
def test_random_logits_through_the_c_abi(seed, scales=None):
    rng, B, T, U, V, blank, lab, il, ll, scale, gs = _draw(seed, scales=scales)
    x = (rng.standard_normal((B, T, U + 1, V)) * scale).astype(np.float32)
    co, go = orc.rnnt_loss_and_grad(x, lab, il, ll, blank, gs)
    run = HipRun(x, lab, il, ll, blank, gs)
    assert run.loss_grad() == 0
    torch.cuda.synchronize()
    assert run.flags() == (0, 0)
    np.testing.assert_allclose(run.costs.cpu().numpy(), co, rtol=COST_RTOL, atol=COST_ATOL, err_msg=str((B, T, U, V, blank)))
    g = run.x.cpu().numpy()
    assert np.abs(g - go).max() < GRAD_ATOL, (B, T, U, V, blank)
    for b in range(B):
        assert not g[b, il[b]:].any() and not g[b, :, ll[b] + 1:].any()


# the first four: draws whose near-zero costs come out 2e-5 .. 4e-3 (relative) off when the log-probabilities are
# formed as x*log2e - round(log-sum-exp)
@pytest.mark.parametrize("seed", [50322, 50384, 50575, 50961, *range(4000, 4020)])
def test_random_logits_with_a_large_spread(seed):
    """Logits of magnitude 30-150 and near-certain paths (costs down to 1e-5): the per-cell log-probabilities must
    be accurate relative to the row maximum, not to the size of the logits."""
    test_random_logits_through_the_c_abi(seed, scales=(10.0, 20.0, 40.0))


@pytest.mark.parametrize("seed", range(2000, 2040))
def test_random_crops_along_t(seed):
    rng, B, T, U, V, blank, lab, il, ll, scale, gs = _draw(seed)
    off = int(rng.integers(0, 4))  # the reference crops 2 frames (util.py:63,101); other offsets share the code
    x = (rng.standard_normal((B, T + off, U + 1, V)) * scale).astype(np.float32)
    co, go = orc.rnnt_loss_and_grad(x[:, off:], lab, il, ll, blank, gs)
    inplace = bool(seed & 1)
    base = torch.from_numpy(x).to(DEV).requires_grad_(True)
    acts = base * 1.0 if inplace else base
    labels = torch.from_numpy(lab).to(DEV) if U else torch.zeros((B, 0), dtype=torch.int32, device=DEV)
    costs = rnnt_amd.rnnt_loss(acts, labels, il, ll, blank, t_offset=off, inplace=inplace)
    (costs * torch.from_numpy(gs).to(DEV)).sum().backward()
    np.testing.assert_allclose(costs.detach().cpu().numpy(), co, rtol=COST_RTOL, atol=COST_ATOL,
                               err_msg=str((B, T, off, U, V, blank)))
    g = base.grad.cpu().numpy()
    assert not g[:, :off].any()
    assert np.abs(g[:, off:] - go).max() < GRAD_ATOL, (B, T, off, U, V, blank)


@pytest.mark.parametrize("seed", range(3000, 3060))
def test_random_additive_joints(seed):
    rng, B, T, U, V, blank, lab, il, ll, scale, gs = _draw(seed, vmax=2500)
    f = (rng.standard_normal((B, T, V)) * scale).astype(np.float32)
    g = (rng.standard_normal((B, U + 1, V)) * scale).astype(np.float32)
    co, gj = orc.rnnt_loss_and_grad(f[:, :, None, :] + g[:, None, :, :], lab, il, ll, blank, gs)
    tf_ = torch.from_numpy(f).to(DEV).requires_grad_(True)
    tg = torch.from_numpy(g).to(DEV).requires_grad_(True)
    costs = rnnt_amd.rnnt_joint_loss(tf_, tg, lab, il, ll, blank)
    (costs * torch.from_numpy(gs).to(DEV)).sum().backward()
    np.testing.assert_allclose(costs.detach().cpu().numpy(), co, rtol=COST_RTOL, atol=COST_ATOL,
                               err_msg=str((B, T, U, V, blank, scale)))
    # d f sums the joint gradient over u, d g over t: up to U+1 (T) terms, each within the bar.  With a symbol repeated
    # a hundred times an element is the difference of two sums of a hundred or more (occupancy times softmax, and
    # the label terms), each accumulated in float32: hence the relative term.  Worst seen in 1.3 * 10^4 draws: 4.9e-5
    # (seed 270381: U = 319, two symbols, logits ~ 6 N(0,1): an element of 16.4 that is the difference of two sums near 55).
    np.testing.assert_allclose(tf_.grad.cpu().numpy(), gj.sum(2), atol=GRAD_ATOL, rtol=SUM_RTOL,
                               err_msg=str((B, T, U, V, blank, scale)))
    np.testing.assert_allclose(tg.grad.cpu().numpy(), gj.sum(1), atol=GRAD_ATOL, rtol=SUM_RTOL,
                               err_msg=str((B, T, U, V, blank, scale)))


@pytest.mark.parametrize("B,T,U,V", [(2, 1, 225, 2), (2, 300, 0, 2), (2, 300, 1, 3), (1, 40, 200, 2), (2, 2, 240, 300)])
def test_additive_joint_with_one_symbol_repeated_throughout(B, T, U, V):
    """Every label the same symbol and (almost) one path through the lattice: the blank / label terms of d f and
    d g are then hundreds of nearly equal numbers, the case in which a float32 running sum drifts furthest."""
    rng = np.random.default_rng(B + T + U + V)
    f = rng.standard_normal((B, T, V)).astype(np.float32)
    g = rng.standard_normal((B, U + 1, V)).astype(np.float32)
    lab = np.zeros((B, U), np.int32)
    il, ll = np.full(B, T, np.int32), np.full(B, U, np.int32)
    gs = np.array([1.7338928, 1.095334], np.float32)[:B]
    blank = V - 1
    co, gj = orc.rnnt_loss_and_grad(f[:, :, None, :] + g[:, None, :, :], lab, il, ll, blank, gs)
    tf_ = torch.from_numpy(f).to(DEV).requires_grad_(True)
    tg = torch.from_numpy(g).to(DEV).requires_grad_(True)
    costs = rnnt_amd.rnnt_joint_loss(tf_, tg, lab, il, ll, blank)
    (costs * torch.from_numpy(gs).to(DEV)).sum().backward()
    np.testing.assert_allclose(costs.detach().cpu().numpy(), co, rtol=COST_RTOL)
    np.testing.assert_allclose(tf_.grad.cpu().numpy(), gj.sum(2), atol=GRAD_ATOL, rtol=COST_RTOL)
    np.testing.assert_allclose(tg.grad.cpu().numpy(), gj.sum(1), atol=GRAD_ATOL, rtol=COST_RTOL)


def _logint(rng, lo, hi):
    return int(round(np.exp(rng.uniform(np.log(lo), np.log(hi + 0.49)))))


@pytest.mark.parametrize("seed", range(5000, 5030))
def test_random_dense_layers(seed):
    """joint_dense (a Dense layer of the reference's get_fc, keras_model.py:83-104) at random widths <= 128."""
    rng = np.random.default_rng(seed)
    M, K, N = _logint(rng, 1, 20000), _logint(rng, 1, 128), _logint(rng, 1, 128)
    act = "tanh" if rng.random() < 0.5 else "linear"
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
    msg = str((M, K, N, act))
    np.testing.assert_allclose(out.detach().cpu().numpy(), y, atol=1e-5, err_msg=msg)
    np.testing.assert_allclose(tx.grad.cpu().numpy(), g @ k.astype(np.float64).T, atol=4e-5 * max(1, N) ** 0.5, err_msg=msg)
    scale = max(1.0, np.sqrt(M))
    np.testing.assert_allclose(tk.grad.cpu().numpy(), x.astype(np.float64).T @ g, atol=2e-5 * scale, err_msg=msg)
    np.testing.assert_allclose(tb.grad.cpu().numpy(), g.sum(0), atol=2e-5 * scale, err_msg=msg)


@pytest.mark.parametrize("seed", range(6000, 6024))
def test_random_fc_joint_heads(seed):
    """concatenate_lambda + get_fc + rnnt_lambda_func_v2 (util.py:76-90, keras_model.py:83-104, util.py:53-74) with
    random widths and depths against the oracle, which builds the tiled tensor literally."""
    rng = np.random.default_rng(seed)
    B, T0, U = int(rng.integers(1, 4)), _logint(rng, 3, 24), _logint(rng, 1, 14) - 1
    He, Hd, V = _logint(rng, 1, 200), _logint(rng, 1, 140), _logint(rng, 2, 90)
    hidden = [_logint(rng, 1, 160) for _ in range(int(rng.integers(0, 4)))]
    blank = int(rng.integers(0, V))
    yt = (rng.standard_normal((B, T0, He)) * 0.5).astype(np.float32)
    yp = (rng.standard_normal((B, U + 1, Hd)) * 0.5).astype(np.float32)
    dims = [He + Hd] + hidden + [V]
    ws = [(rng.uniform(-0.5, 0.5, (dims[i], dims[i + 1])).astype(np.float32) / np.sqrt(max(1.0, dims[i] / 16)).astype(np.float32),
           rng.uniform(-0.1, 0.1, dims[i + 1]).astype(np.float32)) for i in range(len(dims) - 1)]
    lab = rng.integers(0, V - 1, (B, U)).astype(np.int32)
    lab[lab >= blank] += 1
    il = rng.integers(3, T0 + 1, (B, 1)).astype(np.int32)   # lengths before the crop of 2 frames
    ll = rng.integers(0, U + 1, (B, 1)).astype(np.int32)
    il[0], ll[0] = T0, U
    gs = np.full(B, 1.0 / B)
    co, dt, dp, dws = orc.fc_joint_loss_and_grads(yt, yp, ws, lab, il, ll, blank_label=blank, grad_scale=gs)
    t_yt = torch.from_numpy(yt).to(DEV).requires_grad_(True)
    t_yp = torch.from_numpy(yp).to(DEV).requires_grad_(True)
    t_ws = [(torch.from_numpy(k).to(DEV).requires_grad_(True), torch.from_numpy(b).to(DEV).requires_grad_(True)) for k, b in ws]
    labels = torch.from_numpy(lab).to(DEV) if U else torch.zeros((B, 0), dtype=torch.int32, device=DEV)
    costs = rnnt_amd.fc_joint_loss(t_yt, t_yp, t_ws, labels, torch.from_numpy(il).to(DEV), torch.from_numpy(ll).to(DEV),
                                   blank_label=blank)
    costs.mean().backward()
    msg = str((B, T0, U, He, Hd, hidden, V, blank))
    np.testing.assert_allclose(costs.detach().cpu().numpy(), co, rtol=COST_RTOL, atol=COST_ATOL, err_msg=msg)
    np.testing.assert_allclose(t_yt.grad.cpu().numpy(), dt, atol=GRAD_ATOL, err_msg=msg)
    np.testing.assert_allclose(t_yp.grad.cpu().numpy(), dp, atol=GRAD_ATOL, err_msg=msg)
    for (k, b), (dk, db) in zip(t_ws, dws):
        np.testing.assert_allclose(k.grad.cpu().numpy(), dk, atol=GRAD_ATOL, err_msg=msg)
        np.testing.assert_allclose(b.grad.cpu().numpy(), db, atol=GRAD_ATOL, err_msg=msg)


@pytest.mark.parametrize("seed", range(7000, 7030))
def test_random_decoder_steps(seed):
    """Batched joint + Util.softmax(is_log=True) of the decoders (rnnt_decoder.py:156-158,200-202)."""
    rng = np.random.default_rng(seed)
    V, nf, ng, n = _logint(rng, 1, 6000), int(rng.integers(1, 9)), int(rng.integers(1, 9)), int(rng.integers(1, 40))
    scale = float(rng.choice([0.5, 4.0, 20.0]))
    enc = (rng.standard_normal((nf, V)) * scale).astype(np.float32)
    pred = (rng.standard_normal((ng, V)) * scale).astype(np.float32)
    fi, gi = rng.integers(0, nf, n), rng.integers(0, ng, n)
    out = rnnt_amd.joint_log_softmax(torch.from_numpy(enc).to(DEV), torch.from_numpy(pred).to(DEV),
                                     torch.from_numpy(fi).to(DEV), torch.from_numpy(gi).to(DEV)).cpu().numpy()
    for i in range(n):
        # Util.softmax adds in float32 like the kernel does; its float64 restatement is given the float32 sum
        want = orc.util_softmax(enc[fi[i]] + pred[gi[i]], is_log=True)
        np.testing.assert_allclose(out[i], want, atol=1e-5 * max(1.0, scale), err_msg=str((V, nf, ng, n, scale, i)))


@pytest.mark.parametrize("seed", range(8000, 8040))
def test_random_malformed_lengths_and_labels(seed):
    """Out-of-range lengths and labels at random places.  The library's contract: the call succeeds, the validation
    word says what was wrong, an utterance with a bad length gets a NaN cost and a zero gradient, a bad label is
    computed as label 0, and every other utterance is unaffected."""
    rng, B, T, U, V, blank, lab, il, ll, scale, gs = _draw(seed, max_cells=8000)
    x = (rng.standard_normal((B, T, U + 1, V)) * scale).astype(np.float32)
    il_bad, ll_bad, lab_bad = il.copy(), ll.copy(), lab.copy()
    bad_len = np.zeros(B, bool)
    for b in range(B):
        r = rng.random()
        if r < 0.25:
            il_bad[b] = rng.choice([0, -1, T + 1, 2 ** 30, -2 ** 31])
            bad_len[b] = True
        elif r < 0.5:
            ll_bad[b] = rng.choice([-1, U + 1, 2 ** 30, -2 ** 31])
            bad_len[b] = True
    want_flags = 1 if bad_len.any() else 0
    lab_eff = lab.copy()
    for b in range(B):
        if U and rng.random() < 0.4:
            u = int(rng.integers(0, U))
            lab_bad[b, u] = rng.choice([-1, V, V + 7, 2 ** 31 - 1, -2 ** 31])
            lab_eff[b, u] = 0
            if not bad_len[b] and u < ll[b]:   # only labels that are used are looked at
                want_flags |= 2
    run = HipRun(x, lab_bad, il_bad, ll_bad, blank, gs)
    assert run.loss_grad() == 0
    rc, flags = run.flags()
    assert flags == want_flags and rc == (4 if want_flags else 0), (flags, want_flags)
    c, g = run.costs.cpu().numpy(), run.x.cpu().numpy()
    assert np.isnan(c[bad_len]).all() and not g[bad_len].any()
    ok = ~bad_len
    if ok.any():
        co, go = orc.rnnt_loss_and_grad(x[ok], lab_eff[ok], il[ok], ll[ok], blank, gs[ok])
        np.testing.assert_allclose(c[ok], co, rtol=COST_RTOL, atol=COST_ATOL)
        assert np.abs(g[ok] - go).max() < GRAD_ATOL


@pytest.mark.parametrize("seed", range(9000, 9030))
def test_random_malformed_inputs_to_the_fused_joint(seed):
    """The same contract for the fused joint, through the C ABI (the Python layer would reject these on the host)."""
    import ctypes

    from rnnt_amd import _lib

    def p(t):
        return ctypes.c_void_p(t.data_ptr())

    rng, B, T, U, V, blank, lab, il, ll, scale, gs = _draw(seed, max_cells=8000, vmax=700)
    f = (rng.standard_normal((B, T, V)) * scale).astype(np.float32)
    g = (rng.standard_normal((B, U + 1, V)) * scale).astype(np.float32)
    il_bad, ll_bad, lab_bad, lab_eff = il.copy(), ll.copy(), lab.copy(), lab.copy()
    bad_len = np.zeros(B, bool)
    want_flags = 0
    for b in range(B):
        r = rng.random()
        if r < 0.25:
            il_bad[b] = rng.choice([0, -1, T + 1, 2 ** 30, -2 ** 31])
            bad_len[b] = True
        elif r < 0.5:
            ll_bad[b] = rng.choice([-1, U + 1, 2 ** 30, -2 ** 31])
            bad_len[b] = True
        if U and rng.random() < 0.4:
            u = int(rng.integers(0, U))
            lab_bad[b, u] = rng.choice([-1, V, V + 7, 2 ** 31 - 1, -2 ** 31])
            lab_eff[b, u] = 0
            if not bad_len[b] and u < ll[b]:
                want_flags |= 2
    if bad_len.any():
        want_flags |= 1
    L = _lib.lib()
    tf_, tg = torch.from_numpy(f).to(DEV), torch.from_numpy(g).to(DEV)
    tl = torch.from_numpy(lab_bad).to(DEV) if U else torch.zeros(1, dtype=torch.int32, device=DEV)
    til, tll, tgs = (torch.from_numpy(a).to(DEV) for a in (il_bad, ll_bad, gs))
    ws = torch.empty(_lib.joint_workspace_bytes(B, T, U + 1, V), dtype=torch.uint8, device=DEV)
    ws.fill_(0xFF)
    costs = torch.full((B,), 7.0, device=DEV)
    df, dg = torch.full_like(tf_, 7.0), torch.full_like(tg, 7.0)
    assert L.rnnt_mi355x_joint_forward(p(tf_), 0, p(tg), p(tl), p(til), p(tll), B, T, U + 1, V, blank, p(costs), p(ws), None) == 0
    assert L.rnnt_mi355x_joint_backward(p(tf_), 0, p(tg), p(tl), p(til), p(tll), B, T, U + 1, V, blank, p(tgs), p(df), 0,
                                        p(dg), p(ws), None, float("nan")) == 0
    flags = ctypes.c_int32(0)
    rc = L.rnnt_mi355x_read_status(p(ws), ctypes.byref(flags), None)
    assert flags.value == want_flags and rc == (4 if want_flags else 0), (flags.value, want_flags)
    c, dfn, dgn = costs.cpu().numpy(), df.cpu().numpy(), dg.cpu().numpy()
    assert np.isnan(c[bad_len]).all() and not dfn[bad_len].any() and not dgn[bad_len].any()
    ok = ~bad_len
    if ok.any():
        co, gj = orc.rnnt_loss_and_grad((f[:, :, None, :] + g[:, None, :, :])[ok], lab_eff[ok], il[ok], ll[ok], blank, gs[ok])
        np.testing.assert_allclose(c[ok], co, rtol=COST_RTOL, atol=COST_ATOL)
        np.testing.assert_allclose(dfn[ok], gj.sum(2), atol=GRAD_ATOL, rtol=COST_RTOL)
        np.testing.assert_allclose(dgn[ok], gj.sum(1), atol=GRAD_ATOL, rtol=COST_RTOL)
