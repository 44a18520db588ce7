"""The reference-shaped Python surface on the GPU: rnnt_loss autograd, the two
lambda functions (reference rnnt/base/util.py:53-74, 93-117), the module, input
validation, and a sharded run compared with the unsharded one."""
import numpy as np
import pytest
import torch

import rnnt_amd
from oracle import rnnt_oracle as orc
from rnnt_amd.loss import reduce_scalars, _workspace
from tests.hip_runner import make_batch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
COST_RTOL = 1e-5
GRAD_ATOL = 1e-4


def test_autograd_matches_oracle_and_leaves_acts_intact():
    x, lab, il, ll = make_batch(4, 21, 8, 62, 39, seed=3)
    acts = torch.from_numpy(x).to(DEV).requires_grad_(True)
    costs = rnnt_amd.rnnt_loss(acts, torch.from_numpy(lab), torch.from_numpy(il), torch.from_numpy(ll), 39)
    w = torch.tensor([0.5, 1.0, 2.0, 0.25], device=DEV)
    (costs * w).sum().backward()
    co, go = orc.rnnt_loss_and_grad(x, lab, il, ll, 39, w.cpu().numpy())
    np.testing.assert_allclose(costs.detach().cpu().numpy(), co, rtol=COST_RTOL)
    assert np.abs(acts.grad.cpu().numpy() - go).max() < GRAD_ATOL
    assert torch.equal(acts.detach().cpu(), torch.from_numpy(x))


def test_keras_mean_loss_gives_one_over_b_scale():
    """loss = mean_b cost_b, the identity-loss reduction of reference keras_model.py:215-217."""
    x, lab, il, ll = make_batch(3, 10, 4, 16, 15, seed=8)
    acts = torch.from_numpy(x).to(DEV).requires_grad_(True)
    loss = rnnt_amd.RNNTLoss(blank_label=15)(acts, torch.from_numpy(lab), torch.from_numpy(il), torch.from_numpy(ll))
    loss.backward()
    co, go = orc.rnnt_loss_and_grad(x, lab, il, ll, 15, np.full(3, 1 / 3))
    np.testing.assert_allclose(loss.item(), co.mean(), rtol=COST_RTOL)
    assert np.abs(acts.grad.cpu().numpy() - go).max() < GRAD_ATOL


def test_rnnt_lambda_func_additive_joint():
    rng = np.random.default_rng(5)
    B, T0, U, V = 3, 14, 5, 40
    yt = rng.standard_normal((B, T0, V)).astype(np.float32)
    yp = rng.standard_normal((B, U + 1, V)).astype(np.float32)
    lab = rng.integers(0, V - 1, (B, U)).astype(np.int32)
    il = np.array([[14], [9], [12]], np.int32)   # [B,1] as the Keras Inputs deliver them
    ll = np.array([[5], [2], [4]], np.int32)
    t_yt = torch.from_numpy(yt).to(DEV).requires_grad_(True)
    t_yp = torch.from_numpy(yp).to(DEV).requires_grad_(True)
    t_il = torch.from_numpy(il).to(DEV)
    costs = rnnt_amd.rnnt_lambda_func([t_yt, t_yp, torch.from_numpy(lab).to(DEV), t_il, torch.from_numpy(ll).to(DEV)])
    assert costs.shape == (B,)
    costs.mean().backward()
    co, dt, dp = orc.rnnt_lambda_func(yt, yp, lab, il, ll, blank_label=39, grad_scale=np.full(B, 1 / B))
    np.testing.assert_allclose(costs.detach().cpu().numpy(), co, rtol=COST_RTOL)
    assert np.abs(t_yt.grad.cpu().numpy() - dt).max() < GRAD_ATOL
    assert np.abs(t_yp.grad.cpu().numpy() - dp).max() < GRAD_ATOL
    assert not t_yt.grad[:, :2].any()                      # the two cropped frames (util.py:101)
    assert torch.equal(t_il.cpu(), torch.from_numpy(il))   # caller's lengths are not edited in place


def test_rnnt_lambda_func_v2_fc_joint():
    rng = np.random.default_rng(6)
    B, T0, U, V = 2, 11, 4, 62
    a = rng.standard_normal((B, T0, U + 1, V)).astype(np.float32)
    lab = rng.integers(0, 39, (B, U)).astype(np.int32)
    il = np.array([[11], [8]], np.int32)
    ll = np.array([[4], [3]], np.int32)
    t_a = torch.from_numpy(a).to(DEV).requires_grad_(True)
    costs = rnnt_amd.rnnt_lambda_func_v2([t_a, torch.from_numpy(lab).to(DEV), torch.from_numpy(il).to(DEV), torch.from_numpy(ll).to(DEV)])
    costs.sum().backward()
    co, da = orc.rnnt_lambda_func_v2(a, lab, il, ll, blank_label=39)
    np.testing.assert_allclose(costs.detach().cpu().numpy(), co, rtol=COST_RTOL)
    assert np.abs(t_a.grad.cpu().numpy() - da).max() < GRAD_ATOL
    assert torch.equal(t_a.detach().cpu(), torch.from_numpy(a))  # the crop is copied, the joint output survives


def test_inplace_backward_reuses_the_acts_buffer():
    x, lab, il, ll = make_batch(2, 9, 4, 32, 31, seed=12)
    base = torch.from_numpy(x).to(DEV).requires_grad_(True)
    acts = base * 1.0
    ptr = acts.data_ptr()
    costs = rnnt_amd.rnnt_loss(acts, lab, il, ll, 31, inplace=True)
    costs.sum().backward()
    _, go = orc.rnnt_loss_and_grad(x, lab, il, ll, 31)
    assert np.abs(base.grad.cpu().numpy() - go).max() < GRAD_ATOL
    assert acts.data_ptr() == ptr and np.abs(acts.detach().cpu().numpy() - go).max() < GRAD_ATOL


def test_inplace_backward_is_reported_to_autograd():
    """inplace=True overwrites acts through a raw pointer; an op that saved acts for its own backward (tanh saves
    its output) must then get autograd's in-place error, not the gradient's bytes; so must an edit of acts between
    forward and backward, and a second backward."""
    x, lab, il, ll = make_batch(2, 9, 4, 32, 31, seed=14)
    base = torch.from_numpy(x).to(DEV).requires_grad_(True)
    acts = torch.tanh(base)
    costs = rnnt_amd.rnnt_loss(acts, lab, il, ll, 31, inplace=True)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        costs.sum().backward()
    # the same model with inplace=False is fine
    base2 = torch.from_numpy(x).to(DEV).requires_grad_(True)
    rnnt_amd.rnnt_loss(torch.tanh(base2), lab, il, ll, 31).sum().backward()
    _, go = orc.rnnt_loss_and_grad(np.tanh(x.astype(np.float64)), lab, il, ll, 31)
    assert np.abs(base2.grad.cpu().numpy() - go * (1 - np.tanh(x.astype(np.float64)) ** 2)).max() < GRAD_ATOL
    # acts edited between forward and backward
    acts3 = (torch.from_numpy(x).to(DEV).requires_grad_(True)) * 1.0
    c3 = rnnt_amd.rnnt_loss(acts3, lab, il, ll, 31)
    acts3.mul_(2.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        c3.sum().backward()
    # second backward
    acts4 = torch.from_numpy(x).to(DEV).requires_grad_(True)
    c4 = rnnt_amd.rnnt_loss(acts4, lab, il, ll, 31)
    c4.sum().backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="twice"):
        c4.sum().backward()


def test_fused_inplace_call_and_scalars():
    x, lab, il, ll = make_batch(3, 12, 6, 48, 0, seed=13)
    acts = torch.from_numpy(x).to(DEV)
    ws = _workspace(3, 12, 7, acts.device)
    gs = torch.full((3,), 1 / 3)
    costs = rnnt_amd.rnnt_loss_grad_(acts, lab, il, ll, 0, gs, workspace=ws, check_inputs=True)
    co, go = orc.rnnt_loss_and_grad(x, lab, il, ll, 0, gs.numpy())
    np.testing.assert_allclose(costs.cpu().numpy(), co, rtol=COST_RTOL)
    assert np.abs(acts.cpu().numpy() - go).max() < GRAD_ATOL
    s = reduce_scalars(costs, ws).cpu().numpy()
    np.testing.assert_allclose(s, [co.sum(), (go ** 2).sum()], rtol=1e-4)


def test_malformed_inputs_raise():
    x, lab, il, ll = make_batch(2, 6, 3, 8, 7, seed=1, ragged=False)
    acts = torch.from_numpy(x).to(DEV)
    with pytest.raises(ValueError):
        rnnt_amd.rnnt_loss(acts, lab, np.array([7, 6]), ll, 7)                # host lengths: checked on the host
    with pytest.raises(ValueError):
        rnnt_amd.rnnt_loss(acts, lab, torch.tensor([7, 6], device=DEV), ll, 7)  # device lengths: checked on the device
    bad = lab.copy()
    bad[0, 0] = 9
    with pytest.raises(ValueError):
        rnnt_amd.rnnt_loss(acts, bad, il, ll, 7)
    with pytest.raises(ValueError):
        rnnt_amd.rnnt_loss(acts, lab, il, ll, 8)                               # blank outside [0,V)
    with pytest.raises(TypeError):
        rnnt_amd.rnnt_loss(acts.double(), lab, il, ll, 7)
    with pytest.raises(ValueError):
        rnnt_amd.rnnt_loss(acts, lab[:, :2], il, ll, 7)


def test_empty_batch_is_rejected_cleanly():
    acts = torch.zeros(0, 4, 3, 5, device=DEV)
    with pytest.raises(ValueError, match="empty batch"):
        rnnt_amd.rnnt_loss(acts, torch.zeros(0, 2, dtype=torch.int32), torch.zeros(0), torch.zeros(0), 0)


def test_sharded_run_equals_unsharded():
    """Every rank's shard computed one after the other on the one GPU ("fake
    world"): the all-reduced scalars and the per-utterance results must equal
    the unsharded run's."""
    rng = np.random.default_rng(21)
    B, T, U, V, blank = 7, 40, 12, 62, 39
    x = rng.standard_normal((B, T, U + 1, V)).astype(np.float32)
    lab = rng.integers(0, 39, (B, U)).astype(np.int32)
    il = rng.integers(10, T + 1, B).astype(np.int32)
    ll = rng.integers(0, U + 1, B).astype(np.int32)
    full = torch.from_numpy(x).to(DEV)
    ws = _workspace(B, T, U + 1, full.device)
    gs = torch.full((B,), 1.0 / B)
    ref_costs = rnnt_amd.rnnt_loss_grad_(full.clone(), lab, il, ll, blank, gs, workspace=ws)
    ref_s = reduce_scalars(ref_costs, ws).cpu().numpy()
    ref_grad = full.clone()
    rnnt_amd.rnnt_loss_grad_(ref_grad, lab, il, ll, blank, gs)
    total = np.zeros(2)
    for world in (2, 3):
        total[:] = 0
        for rank in range(world):
            a, lb, i2, l2, idx = rnnt_amd.shard_batch(torch.from_numpy(x).to(DEV), torch.from_numpy(lab), il, ll, rank, world)
            w2 = _workspace(len(idx), a.shape[1], a.shape[2], a.device)
            c = rnnt_amd.rnnt_loss_grad_(a, lb, i2, l2, blank, torch.full((len(idx),), 1.0 / B), workspace=w2)
            total += reduce_scalars(c, w2).cpu().numpy()
            for k, i in enumerate(idx):
                assert torch.equal(c[k], ref_costs[i])
                assert torch.equal(a[k, : il[i], : ll[i] + 1], ref_grad[i, : il[i], : ll[i] + 1])
        np.testing.assert_allclose(total, ref_s, rtol=1e-6)


def test_t_offset_crop_without_copy_matches_explicit_crop():
    """rnnt_loss(acts, ..., t_offset=2) == rnnt_loss(acts[:, 2:].contiguous(), ...) and the skipped
    frames get a zero gradient (the crop of reference util.py:62-64)."""
    for V, blank in ((256, 255), (62, 39)):   # generic kernels and the small-vocabulary path (falls back: strided)
        x, lab, il, ll = make_batch(3, 13, 5, V, blank, seed=V)
        il = np.maximum(il - 2, 1).astype(np.int32)
        a = torch.from_numpy(x).to(DEV).requires_grad_(True)
        costs = rnnt_amd.rnnt_loss(a, lab, il, ll, blank, t_offset=2)
        costs.sum().backward()
        co, go = orc.rnnt_loss_and_grad(x[:, 2:], lab, il, ll, blank)
        np.testing.assert_allclose(costs.detach().cpu().numpy(), co, rtol=COST_RTOL)
        g = a.grad.cpu().numpy()
        assert not g[:, :2].any()
        assert np.abs(g[:, 2:] - go).max() < GRAD_ATOL
        assert torch.equal(a.detach().cpu(), torch.from_numpy(x))


def test_non_contiguous_acts_are_accepted():
    x, lab, il, ll = make_batch(2, 7, 3, 24, 23, seed=2)
    base = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1, 3))).to(DEV).requires_grad_(True)  # [B,U1,T,V]
    acts = base.permute(0, 2, 1, 3)                                                                      # view [B,T,U1,V]
    assert not acts.is_contiguous()
    costs = rnnt_amd.rnnt_loss(acts, lab, il, ll, 23)
    costs.sum().backward()
    co, go = orc.rnnt_loss_and_grad(x, lab, il, ll, 23)
    np.testing.assert_allclose(costs.detach().cpu().numpy(), co, rtol=COST_RTOL)
    assert np.abs(base.grad.permute(0, 2, 1, 3).cpu().numpy() - go).max() < GRAD_ATOL


JOINT_SHAPES = [  # B, T, U, V, blank -> every elements-per-lane variant of the fused joint kernels
    (3, 9, 4, 40, 39), (2, 11, 5, 62, 39), (2, 8, 3, 100, 0), (2, 7, 4, 200, 199), (2, 9, 3, 512, 5),
    (2, 6, 4, 1000, 999), (2, 5, 2, 1024, 1023), (1, 4, 3, 2000, 7), (1, 3, 2, 4096, 4095), (2, 1, 3, 30, 2),
    (2, 6, 0, 30, 2), (1, 20, 130, 16, 15),
    # several 32-row tiles / column strips of the contractions, ragged edges in every dimension
    (2, 70, 40, 150, 149), (1, 45, 70, 33, 0), (2, 33, 64, 129, 64), (1, 100, 12, 300, 17),
    (1, 6, 600, 9, 8),   # more than 512 label positions
    # few wavefront tiles in the d g contraction: its sum over frames is cut into 4 / 8 slices
    (1, 300, 20, 62, 39), (2, 520, 5, 300, 7), (3, 130, 33, 40, 0),
]


@pytest.mark.parametrize("B,T,U,V,blank", JOINT_SHAPES)
def test_fused_joint_matches_oracle(B, T, U, V, blank):
    rng = np.random.default_rng(B * 7 + T * 5 + U * 3 + V)
    f = rng.standard_normal((B, T, V)).astype(np.float32) * 1.5
    g = rng.standard_normal((B, U + 1, V)).astype(np.float32) * 1.5
    lab = rng.integers(0, V - 1, (B, U)).astype(np.int32)
    lab[lab >= blank] += 1
    il = rng.integers(max(1, T // 2), T + 1, B).astype(np.int32)
    ll = rng.integers(0, U + 1, B).astype(np.int32)
    il[0], ll[0] = T, U
    w = np.linspace(0.5, 1.5, B).astype(np.float32)
    joint = f[:, :, None, :] + g[:, None, :, :]
    co, gj = orc.rnnt_loss_and_grad(joint, lab, il, ll, blank, w)
    tf_ = torch.from_numpy(f).to(DEV).requires_grad_(True)
    tg = torch.from_numpy(g).to(DEV).requires_grad_(True)
    costs = rnnt_amd.rnnt_joint_loss(tf_, tg, lab, il, ll, blank)
    (costs * torch.from_numpy(w).to(DEV)).sum().backward()
    np.testing.assert_allclose(costs.detach().cpu().numpy(), co, rtol=COST_RTOL)
    # d f = sum over u, d g = sum over t of the joint gradient: up to U+1 (T) terms, each within the 1e-4 bar; sums
    # of hundreds of terms reach tens to hundreds, where float32 resolves ~1e-5 of the value: hence the relative term
    np.testing.assert_allclose(tf_.grad.cpu().numpy(), gj.sum(2), atol=GRAD_ATOL, rtol=COST_RTOL)
    np.testing.assert_allclose(tg.grad.cpu().numpy(), gj.sum(1), atol=GRAD_ATOL, rtol=COST_RTOL)


def test_fused_joint_equals_materialised_path_and_handles_separated_maxima():
    """f and g peak at different vocabulary entries with a large margin: the sum of
    products 2^(f-mf)*2^(g-mg) underflows and the exact-maximum fallback must take over."""
    B, T, U, V, blank = 2, 6, 3, 96, 95
    rng = np.random.default_rng(4)
    f = rng.standard_normal((B, T, V)).astype(np.float32)
    g = rng.standard_normal((B, U + 1, V)).astype(np.float32)
    f[:, :, 3] += 400.0          # f's maximum
    g[:, :, 70] += 400.0         # g's maximum elsewhere: products of the two normalised terms vanish
    lab = rng.integers(0, V - 1, (B, U)).astype(np.int32)
    il, ll = np.array([T, T - 2], np.int32), np.array([U, 1], np.int32)
    joint = f[:, :, None, :] + g[:, None, :, :]
    co, gj = orc.rnnt_loss_and_grad(joint, lab, il, ll, blank)
    tf_ = torch.from_numpy(f).to(DEV).requires_grad_(True)
    tg = torch.from_numpy(g).to(DEV).requires_grad_(True)
    costs = rnnt_amd.rnnt_joint_loss(tf_, tg, lab, il, ll, blank)
    costs.sum().backward()
    np.testing.assert_allclose(costs.detach().cpu().numpy(), co, rtol=COST_RTOL)
    assert np.abs(tf_.grad.cpu().numpy() - gj.sum(2)).max() < GRAD_ATOL
    assert np.abs(tg.grad.cpu().numpy() - gj.sum(1)).max() < GRAD_ATOL


def test_fused_joint_with_some_cells_below_the_linear_domain_floor():
    """Only part of the lattice has separated maxima, by a margin (60 nats = 2^-87)
    that leaves the linear-domain sum representable but below the trusted floor:
    flagged and unflagged cells then share rows of d f and d g."""
    B, T, U, V, blank = 2, 40, 9, 70, 69
    rng = np.random.default_rng(41)
    f = rng.standard_normal((B, T, V)).astype(np.float32)
    g = rng.standard_normal((B, U + 1, V)).astype(np.float32)
    f[:, 5:23:2, 3] += 60.0
    g[:, 2:7, 50] += 60.0
    lab = rng.integers(0, V - 1, (B, U)).astype(np.int32)
    il, ll = np.array([T, T - 7], np.int32), np.array([U, 6], np.int32)
    joint = f[:, :, None, :] + g[:, None, :, :]
    w = np.array([0.7, 1.3], np.float32)
    co, gj = orc.rnnt_loss_and_grad(joint, lab, il, ll, blank, w)
    tf_ = torch.from_numpy(f).to(DEV).requires_grad_(True)
    tg = torch.from_numpy(g).to(DEV).requires_grad_(True)
    costs = rnnt_amd.rnnt_joint_loss(tf_, tg, lab, il, ll, blank)
    (costs * torch.from_numpy(w).to(DEV)).sum().backward()
    np.testing.assert_allclose(costs.detach().cpu().numpy(), co, rtol=COST_RTOL)
    assert np.abs(tf_.grad.cpu().numpy() - gj.sum(2)).max() < GRAD_ATOL
    assert np.abs(tg.grad.cpu().numpy() - gj.sum(1)).max() < GRAD_ATOL


def test_fused_joint_is_deterministic_with_repeated_labels():
    B, T, U, V, blank = 2, 37, 20, 12, 3   # 20 labels over 11 symbols: many repeats; blank mid-vocabulary
    rng = np.random.default_rng(43)
    f = rng.standard_normal((B, T, V)).astype(np.float32)
    g = rng.standard_normal((B, U + 1, V)).astype(np.float32)
    lab = rng.integers(0, V - 1, (B, U)).astype(np.int32)
    lab[lab >= blank] += 1
    il, ll = np.array([T, 30], np.int32), np.array([U, 13], np.int32)
    joint = f[:, :, None, :] + g[:, None, :, :]
    co, gj = orc.rnnt_loss_and_grad(joint, lab, il, ll, blank)
    outs = []
    for _ in range(3):
        tf_ = torch.from_numpy(f).to(DEV).requires_grad_(True)
        tg = torch.from_numpy(g).to(DEV).requires_grad_(True)
        rnnt_amd.rnnt_joint_loss(tf_, tg, lab, il, ll, blank).sum().backward()
        outs.append((tf_.grad.clone(), tg.grad.clone()))
    assert np.abs(outs[0][0].cpu().numpy() - gj.sum(2)).max() < GRAD_ATOL
    assert np.abs(outs[0][1].cpu().numpy() - gj.sum(1)).max() < GRAD_ATOL
    for a, b in outs[1:]:
        assert torch.equal(a, outs[0][0]) and torch.equal(b, outs[0][1])


def test_lambda_func_fused_and_unfused_agree():
    rng = np.random.default_rng(15)
    B, T0, U, V = 3, 16, 6, 62
    yt = rng.standard_normal((B, T0, V)).astype(np.float32)
    yp = rng.standard_normal((B, U + 1, V)).astype(np.float32)
    lab = rng.integers(0, 39, (B, U)).astype(np.int32)
    il = torch.tensor([[16], [11], [14]], dtype=torch.int32, device=DEV)
    ll = torch.tensor([[6], [3], [5]], dtype=torch.int32, device=DEV)
    res = []
    for fused in (True, False):
        a = torch.from_numpy(yt).to(DEV).requires_grad_(True)
        b = torch.from_numpy(yp).to(DEV).requires_grad_(True)
        c = rnnt_amd.rnnt_lambda_func([a, b, torch.from_numpy(lab).to(DEV), il, ll], fused=fused)
        c.mean().backward()
        res.append((c.detach().cpu().numpy(), a.grad.cpu().numpy(), b.grad.cpu().numpy()))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=COST_RTOL)
    assert np.abs(res[0][1] - res[1][1]).max() < GRAD_ATOL and np.abs(res[0][2] - res[1][2]).max() < GRAD_ATOL
    assert not res[0][1][:, :2].any()


def test_entry_points_can_be_captured_in_a_hip_graph():
    """No entry point allocates, frees or synchronises, so a whole loss + gradient step can be recorded
    once into a HIP graph and replayed (what a launch-bound small-batch training loop wants)."""
    import ctypes
    from rnnt_amd import _lib
    x, lab, il, ll = make_batch(4, 30, 12, 62, 39, seed=21)
    co, go = orc.rnnt_loss_and_grad(x, lab, il, ll, 39, np.full(4, 0.25))
    dev = torch.device(DEV)
    master = torch.from_numpy(x).to(dev)
    buf = master.clone()
    t_lab, t_il, t_ll = (torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev) for a in (lab, il, ll))
    gs = torch.full((4,), 0.25, device=dev)
    costs = torch.zeros(4, device=dev)
    scal = torch.zeros(2, dtype=torch.float64, device=dev)
    ws = torch.empty(_lib.workspace_bytes(4, 30, 13), dtype=torch.uint8, device=dev)
    L = _lib.lib()
    p = lambda t: ctypes.c_void_p(t.data_ptr())

    def step(stream):
        sp = ctypes.c_void_p(stream.cuda_stream)
        buf.copy_(master)
        assert L.rnnt_mi355x_loss_grad(p(buf), p(t_lab), p(t_il), p(t_ll), 4, 30, 13, 62, 39, p(gs), p(costs), p(ws), sp,
                                       float("nan")) == 0
        assert L.rnnt_mi355x_reduce_scalars(p(costs), 4, p(ws), p(scal), sp) == 0

    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        step(side)                                  # warm-up outside the capture
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step(torch.cuda.current_stream(dev))
    for _ in range(3):
        costs.zero_()
        graph.replay()
    torch.cuda.synchronize()
    np.testing.assert_allclose(costs.cpu().numpy(), co, rtol=COST_RTOL)
    assert np.abs(buf.cpu().numpy() - go).max() < GRAD_ATOL
    np.testing.assert_allclose(scal.cpu().numpy()[0], co.sum(), rtol=COST_RTOL)


# B, T0, t_offset, U, V, blank: small-vocabulary tiles (64 rows) with cropped frames inside full tiles, inside the
# last partial tile only, a large vocabulary (generic kernels + zero-fill of the cropped frames), and crops that
# skip nearly the whole tensor (more tiles than the workspace has per-block slots for the lattice that is left:
# the gradient pass then takes its persistent form)
CROP_SHAPES = [(2, 3, 2, 0, 5, 0), (3, 9, 2, 4, 62, 39), (2, 40, 3, 7, 40, 39), (1, 5, 4, 2, 16, 1), (2, 7, 2, 3, 300, 7),
               (2, 6, 1, 3, 1024, 0), (1, 70, 68, 20, 62, 39), (2, 90, 87, 9, 50, 3)]


@pytest.mark.parametrize("B,T0,off,U,V,blank", CROP_SHAPES)
@pytest.mark.parametrize("inplace", [False, True])
def test_crop_along_t_is_done_inside_the_library(B, T0, off, U, V, blank, inplace):
    rng = np.random.default_rng(B + T0 + U + V)
    x = rng.standard_normal((B, T0, U + 1, V)).astype(np.float32)
    lab = rng.integers(0, V - 1, (B, max(U, 1))).astype(np.int32)[:, :U]
    lab[lab >= blank] += 1
    T = T0 - off
    il = rng.integers(1, T + 1, B).astype(np.int32)
    ll = rng.integers(0, U + 1, B).astype(np.int32)
    il[0], ll[0] = T, U
    co, go = orc.rnnt_loss_and_grad(x[:, off:], lab if U else np.zeros((B, 0), np.int32), il, ll, blank)
    base = torch.from_numpy(x).to(DEV).requires_grad_(True)
    acts = base * 1.0 if inplace else base
    labels = torch.from_numpy(lab).to(DEV) if U else torch.zeros((B, 0), dtype=torch.int32, device=DEV)
    costs = rnnt_amd.rnnt_loss(acts, labels, il, ll, blank, t_offset=off, inplace=inplace)
    costs.sum().backward()
    np.testing.assert_allclose(costs.detach().cpu().numpy(), co, rtol=COST_RTOL)
    g = base.grad.cpu().numpy()
    assert not g[:, :off].any()                      # the cropped frames (util.py:63)
    assert np.abs(g[:, off:] - go).max() < GRAD_ATOL


@pytest.mark.parametrize("scale", [8.0, 25.0])
def test_fused_joint_with_peaked_outputs(scale):
    """Trained networks emit logits with a large spread: most of exp(f - max f) and exp(g - max g) is tiny or
    zero, some cells fall under the linear-domain floor and take the exponent-domain route."""
    B, T, U, V, blank = 2, 45, 9, 62, 39
    rng = np.random.default_rng(int(scale))
    f = (rng.standard_normal((B, T, V)) * scale).astype(np.float32)
    g = (rng.standard_normal((B, U + 1, V)) * scale).astype(np.float32)
    lab = rng.integers(0, V - 1, (B, U)).astype(np.int32)
    lab[lab >= blank] += 1
    il, ll = np.array([T, T - 11], np.int32), np.array([U, 5], np.int32)
    joint = f[:, :, None, :].astype(np.float64) + g[:, None, :, :].astype(np.float64)
    co, gj = orc.rnnt_loss_and_grad(joint, lab, il, ll, blank)
    tf_ = torch.from_numpy(f).to(DEV).requires_grad_(True)
    tg = torch.from_numpy(g).to(DEV).requires_grad_(True)
    costs = rnnt_amd.rnnt_joint_loss(tf_, tg, lab, il, ll, blank)
    costs.sum().backward()
    np.testing.assert_allclose(costs.detach().cpu().numpy(), co, rtol=COST_RTOL)
    assert np.abs(tf_.grad.cpu().numpy() - gj.sum(2)).max() < GRAD_ATOL
    assert np.abs(tg.grad.cpu().numpy() - gj.sum(1)).max() < GRAD_ATOL


def test_fused_joint_with_masked_vocabulary_entries():
    B, T, U, V, blank = 2, 20, 5, 70, 69
    rng = np.random.default_rng(91)
    f = rng.standard_normal((B, T, V)).astype(np.float32)
    g = rng.standard_normal((B, U + 1, V)).astype(np.float32)
    lab = rng.integers(0, V - 1, (B, U)).astype(np.int32)
    masked = np.setdiff1d(np.arange(V - 1), lab.reshape(-1))[:12]   # symbols no utterance needs
    f[:, ::3, masked[:6]] = -np.inf
    g[:, 1::2, masked[6:]] = -np.inf
    il, ll = np.array([T, T - 3], np.int32), np.array([U, 3], np.int32)
    with np.errstate(invalid="ignore"):
        joint = f[:, :, None, :].astype(np.float64) + g[:, None, :, :].astype(np.float64)
        co, gj = orc.rnnt_loss_and_grad(joint, lab, il, ll, blank)
    tf_ = torch.from_numpy(f).to(DEV).requires_grad_(True)
    tg = torch.from_numpy(g).to(DEV).requires_grad_(True)
    costs = rnnt_amd.rnnt_joint_loss(tf_, tg, lab, il, ll, blank)
    costs.sum().backward()
    np.testing.assert_allclose(costs.detach().cpu().numpy(), co, rtol=COST_RTOL)
    df, dg = tf_.grad.cpu().numpy(), tg.grad.cpu().numpy()
    assert np.isfinite(df).all() and np.isfinite(dg).all()
    assert np.abs(df - gj.sum(2)).max() < GRAD_ATOL and np.abs(dg - gj.sum(1)).max() < GRAD_ATOL


def test_batch_producer_arrays_feed_the_lambda():
    """pad_batch's arrays (float64 lengths of shape [B], float labels padded with 0: data_generator.py:391-426) go
    through rnnt_lambda_func as they are -- the casts the Keras Inputs do (keras_model.py:125-127) happen inside."""
    rng = np.random.default_rng(12)
    V, F = 41, 6
    feats = [rng.standard_normal((t, F)) for t in (13, 9, 11)]
    seqs = [[3, 7, 7, 1], [5], [2, 0, 9]]
    inputs, _ = rnnt_amd.pad_batch(feats, seqs, vocab_size=V - 1)
    We, Wd = rng.standard_normal((F, V)) * 0.3, rng.standard_normal((V - 1, V)) * 0.3
    yt = (inputs["INPUT_TRANS"] @ We).astype(np.float32)          # stand-ins for the two networks
    yp = (inputs["INPUT_PREDS"] @ Wd).astype(np.float32)
    t_yt = torch.from_numpy(yt).to(DEV).requires_grad_(True)
    t_yp = torch.from_numpy(yp).to(DEV).requires_grad_(True)
    costs = rnnt_amd.rnnt_lambda_func([t_yt, t_yp, inputs["INPUT_LABEL"], inputs["INPUT_INLEN"], inputs["INPUT_LBLEN"]],
                                      blank_label=V - 1)
    costs.mean().backward()
    il = inputs["INPUT_INLEN"].astype(np.int32).reshape(-1, 1)
    ll = inputs["INPUT_LBLEN"].astype(np.int32).reshape(-1, 1)
    co, dt, dp = orc.rnnt_lambda_func(yt, yp, inputs["INPUT_LABEL"].astype(np.int32), il, ll, blank_label=V - 1,
                                      grad_scale=np.full(3, 1 / 3))
    np.testing.assert_allclose(costs.detach().cpu().numpy(), co, rtol=COST_RTOL)
    assert np.abs(t_yt.grad.cpu().numpy() - dt).max() < GRAD_ATOL
    assert np.abs(t_yp.grad.cpu().numpy() - dp).max() < GRAD_ATOL


def test_loss_grad_replays_a_graph_for_a_caller_that_keeps_its_buffers():
    """Small shapes with static buffers: the third call with the same addresses is a graph replay and gives the very
    bits of the eager call; new addresses run eagerly again; graph=True insists and refuses host lengths."""
    import rnnt_amd
    from rnnt_amd import loss as loss_mod
    B, T, U, V, blank = 4, 30, 9, 62, 39
    x, lab, il, ll = make_batch(B, T, U, V, blank, seed=77)
    acts0 = torch.from_numpy(x).to(DEV)
    labd = torch.from_numpy(lab).to(DEV)
    ild, lld = torch.from_numpy(il).to(DEV), torch.from_numpy(ll).to(DEV)
    gs = torch.full((B,), 0.25, device=DEV)
    ws = torch.empty(rnnt_amd._lib.workspace_bytes(B, T, U + 1), dtype=torch.uint8, device=DEV)
    loss_mod._GRAPH_CACHE.clear(); loss_mod._GRAPH_SEEN.clear()
    eager_acts = acts0.clone()
    eager_costs = rnnt_amd.rnnt_loss_grad_(eager_acts, labd, ild, lld, blank, gs, workspace=ws, graph=False).clone()
    buf = torch.empty_like(acts0)
    for i in range(4):
        buf.copy_(acts0)
        costs = rnnt_amd.rnnt_loss_grad_(buf, labd, ild, lld, blank, gs, workspace=ws)
        torch.cuda.synchronize()
        assert torch.equal(costs, eager_costs) and torch.equal(buf, eager_acts), i
        assert len(loss_mod._GRAPH_CACHE) == (0 if i == 0 else 1)
    other = acts0.clone()   # other addresses: eager again, no new graph yet
    c2 = rnnt_amd.rnnt_loss_grad_(other, labd, ild, lld, blank, gs, workspace=ws)
    assert torch.equal(c2, eager_costs) and len(loss_mod._GRAPH_CACHE) == 1
    with pytest.raises(ValueError, match="graph=True"):
        rnnt_amd.rnnt_loss_grad_(acts0.clone(), labd, il, ll, blank, gs, workspace=ws, graph=True)
    loss_mod._GRAPH_CACHE.clear(); loss_mod._GRAPH_SEEN.clear()
