"""BASELINE-sized runs on the GPU, checked through size-independent
properties (the float64 oracle cannot afford the whole tensor) plus exact
oracle parity on a sampled subset of utterances.

Properties used:
  * every gradient row sums to zero (softmax backward of a function of the
    log-probabilities);
  * identical utterances give bit-identical costs and gradients wherever they
    sit in the batch (also beyond the 2^31-element mark of the logits tensor);
  * cells beyond an utterance's lengths are exact zeros;
  * whatever the gradient kernel writes as zero without computing it is negligible in the exact gradient.
"""
import numpy as np
import pytest
import torch

from oracle import rnnt_oracle as orc
from tests.hip_runner import HipRun

pytestmark = pytest.mark.gpu
COST_RTOL = 1e-5
GRAD_ATOL = 1e-4


def _gpu_batch(B, T, U, V, blank, seed, il=None, ll=None):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.empty((B, T, U + 1, V), dtype=torch.float32, device=dev).normal_(generator=g)
    lab = torch.randint(0, V - 1, (B, U), generator=g, device=dev, dtype=torch.int32)
    lab = lab + (lab >= blank).to(torch.int32)
    il = np.full(B, T, np.int32) if il is None else il
    ll = np.full(B, U, np.int32) if ll is None else ll
    return x, lab, il, ll


def _run(x, lab, il, ll, blank, gs=None, prune_log2=None, stream=None):
    run = HipRun.__new__(HipRun)
    import ctypes
    from rnnt_amd import _lib
    run.dev = x.device
    run.prune = _lib.prune_arg(prune_log2)
    run.B, run.T, run.U1, run.V = x.shape
    run.blank = blank
    run.x, run.lab = x, lab
    run.il = torch.from_numpy(il).to(x.device)
    run.ll = torch.from_numpy(ll).to(x.device)
    run.gs = gs
    run.ws = torch.empty(_lib.workspace_bytes(run.B, run.T, run.U1), dtype=torch.uint8, device=x.device)
    run.costs = torch.full((run.B,), float("nan"), dtype=torch.float32, device=x.device)
    run.stream = ctypes.c_void_p((stream or torch.cuda.current_stream(x.device)).cuda_stream)
    assert run.loss_grad() == 0
    if stream is not None:  # the caller waits for its streams itself
        return run
    torch.cuda.synchronize()
    assert run.flags() == (0, 0)
    return run


def _subset_parity(x_before, run, lab, il, ll, blank, picks, scale=1.0):
    for b in picks:
        T, U = int(il[b]), int(ll[b])
        xb = x_before[b:b + 1, :T, :U + 1].cpu().numpy()
        co, go = orc.rnnt_loss_and_grad(xb, lab[b:b + 1, :U].cpu().numpy(), [T], [U], blank, [scale])
        np.testing.assert_allclose(run.costs[b].item(), co[0], rtol=COST_RTOL)
        got = run.x[b, :T, :U + 1].cpu().numpy()
        assert np.abs(got - go).max() < GRAD_ATOL


def test_config3_full_size_properties_and_subset_parity():
    """B=64 T=400 U=100 V=1024 (10.6 GB of logits; 2.65e9 elements > 2^31)."""
    B, T, U, V, blank = 64, 400, 100, 1024, 1023
    x, lab, il, ll = _gpu_batch(B, T, U, V, blank, seed=0xC0FFEE + 3)
    # utterances 5 and 61 are copies of utterance 0 (one before, one beyond the 2^31-element mark)
    x[5].copy_(x[0]); x[61].copy_(x[0]); lab[5].copy_(lab[0]); lab[61].copy_(lab[0])
    keep = {b: x[b].clone() for b in (0, 63)}
    run = _run(x, lab, il, ll, blank)
    assert torch.isfinite(run.costs).all()
    assert torch.equal(run.costs[5], run.costs[0]) and torch.equal(run.costs[61], run.costs[0])
    assert torch.equal(run.x[5], run.x[0]) and torch.equal(run.x[61], run.x[0])
    rows = run.x.sum(-1)
    assert rows.abs().max().item() < 2e-5          # every gradient row sums to ~0
    assert torch.isfinite(run.x[63]).all()
    before = torch.stack([keep[0], keep[63]])
    sub = HipRun.__new__(HipRun)
    sub.costs = run.costs[[0, 63]]
    sub.x = run.x[[0, 63]]
    _subset_parity(before, sub, lab[[0, 63]], il[[0, 63]], ll[[0, 63]], blank, [0, 1])


def test_longest_lattice_of_128_columns_closed_form():
    """One utterance at the edge of the shape limit ((maxT + maxU1 + 32) * maxU1 < 2^27: the sweeps' byte offsets into
    the records end just below 2^31).  Constant logits make every path equally likely, so the cost is known in closed
    form: -ln C(T-1+U, U) + (T+U) ln V."""
    import math
    T, U, V, blank = (1 << 20) - 161, 127, 4, 0
    assert (T + U + 1 + 32) * (U + 1) < (1 << 27) <= (T + 1 + U + 1 + 32) * (U + 1)
    dev = torch.device("cuda:0")
    x = torch.zeros((1, T, U + 1, V), dtype=torch.float32, device=dev)
    lab = torch.ones((1, U), dtype=torch.int32, device=dev)
    run = _run(x, lab, np.array([T], np.int32), np.array([U], np.int32), blank)
    log_paths = math.lgamma(T + U) - math.lgamma(U + 1) - math.lgamma(T)
    want = -log_paths + (T + U) * math.log(V)
    assert abs(run.costs[0].item() - want) < 1e-6 * want
    assert run.x.sum(-1).abs().max().item() < 2e-5
    # the probability of passing through a cell, summed over an anti-diagonal's blank and label moves, is 1:
    # minus the gradient at the blank and the label column, summed over the first frame's cells and the moves out of
    # (0, 0), is what leaves the start
    g00 = run.x[0, 0, 0]
    assert abs(-(g00[blank] + g00[1]).item() + 2 * 0.25 - 1.0) < 1e-4  # dx = gamma*softmax - moves; gamma(0,0) = 1


def test_config2_full_size_subset_parity():
    """B=32 T=500 U=100 V=62, blank=39 (reference util.py:72)."""
    B, T, U, V, blank = 32, 500, 100, 62, 39
    x, lab, il, ll = _gpu_batch(B, T, U, V, blank, seed=0xC0FFEE + 2)
    before = x[[0, 17, 31]].clone()
    run = _run(x, lab, il, ll, blank)
    assert run.x.sum(-1).abs().max().item() < 2e-5
    sub = HipRun.__new__(HipRun)
    sub.costs, sub.x = run.costs[[0, 17, 31]], run.x[[0, 17, 31]]
    _subset_parity(before, sub, lab[[0, 17, 31]], il[[0, 17, 31]], ll[[0, 17, 31]], blank, [0, 1, 2])


def test_config4_vocab_reduced_batch():
    """V=4096, T=250, U=150 with B=4 (the per-GPU share of config 4 on 8 GPUs)."""
    B, T, U, V, blank = 4, 250, 150, 4096, 4095
    x, lab, il, ll = _gpu_batch(B, T, U, V, blank, seed=0xC0FFEE + 4)
    before = x[[1]].clone()
    gs = torch.full((B,), 1.0 / 32, device=x.device)
    run = _run(x, lab, il, ll, blank, gs)
    assert run.x.sum(-1).abs().max().item() < 1e-6
    sub = HipRun.__new__(HipRun)
    sub.costs, sub.x = run.costs[[1]], run.x[[1]]
    _subset_parity(before, sub, lab[[1]], il[[1]], ll[[1]], blank, [0], scale=1.0 / 32)


def test_config4_full_global_batch():
    """Config 4 whole: B=32 T=250 U=150 V=4096, 4.95e9 elements (19.8 GB): the second tensor beyond 2^31 elements,
    rows of 16 KiB (16 chunks per lane)."""
    B, T, U, V, blank = 32, 250, 150, 4096, 4095
    x, lab, il, ll = _gpu_batch(B, T, U, V, blank, seed=0xC0FFEE + 4)
    # copies of utterance 0 just before, just beyond and far beyond the 2^31-element mark (13.9 utterances)
    for b in (13, 14, 31):
        x[b].copy_(x[0]); lab[b].copy_(lab[0])
    picks = [0, 17, 30]
    before = x[picks].clone()
    gs = torch.full((B,), 1.0 / B, device=x.device)
    run = _run(x, lab, il, ll, blank, gs)
    assert torch.isfinite(run.costs).all()
    for b in (13, 14, 31):
        assert torch.equal(run.costs[b], run.costs[0]) and torch.equal(run.x[b], run.x[0])
    for b in range(B):  # (one utterance at a time: the row sums of the whole tensor would double its footprint)
        assert run.x[b].sum(-1).abs().max().item() < 1e-6
    sub = HipRun.__new__(HipRun)
    sub.costs, sub.x = run.costs[picks], run.x[picks]
    _subset_parity(before, sub, lab[picks], il[picks], ll[picks], blank, [0, 1, 2], scale=1.0 / B)


def test_config5_full_global_batch_padded_to_the_maxima():
    """Config 5 whole on one GPU: B=128, T in [50,1000], U in [10,200], V=512, padded to 1000 x 201 (52.7 GB,
    1.3e10 elements): exact zeros beyond every utterance's lengths, duplicates bit-equal across the 2^31 mark,
    oracle parity on the longest, the shortest and one more utterance."""
    B, T, U, V, blank = 128, 1000, 200, 512, 511
    rng = np.random.default_rng(0xC0FFEE + 5)
    il = rng.integers(50, T + 1, B).astype(np.int32)
    ll = rng.integers(10, U + 1, B).astype(np.int32)
    il[7], ll[7] = T, U
    il[90], ll[90] = 50, 10
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(0xC0FFEE + 5)
    x = torch.empty((B, T, U + 1, V), dtype=torch.float32, device=dev)
    for b in range(B):
        x[b].normal_(generator=g)
    lab = torch.randint(0, V - 1, (B, U), generator=g, device=dev, dtype=torch.int32)
    # utterance 100 (beyond the 2^31-element mark, which falls inside utterance 20) repeats utterance 3
    x[100].copy_(x[3]); lab[100].copy_(lab[3]); il[100], ll[100] = il[3], ll[3]
    longest = int(np.argmax(il.astype(np.int64) * (ll + 1)))
    shortest = int(np.argmin(il.astype(np.int64) * (ll + 1)))
    picks = sorted({longest, shortest, 64})
    before = x[picks].clone()
    gs = torch.full((B,), 1.0 / B, device=dev)
    run = _run(x, lab, il, ll, blank, gs)
    assert torch.isfinite(run.costs).all()
    assert torch.equal(run.costs[100], run.costs[3]) and torch.equal(run.x[100], run.x[3])
    for b in range(B):
        assert not run.x[b, il[b]:].any() and not run.x[b, :, ll[b] + 1:].any()
        assert run.x[b].sum(-1).abs().max().item() < 1e-6
    sub = HipRun.__new__(HipRun)
    sub.costs, sub.x = run.costs[picks], run.x[picks]
    _subset_parity(before, sub, lab[picks], il[picks], ll[picks], blank, list(range(len(picks))), scale=1.0 / B)
    del run, x, before
    torch.cuda.empty_cache()


def test_config5_ragged_padded_cells_and_subset_parity():
    """Ragged lengths T in [50,1000], U in [10,200], V=512, B=16 of the 128."""
    B, T, U, V, blank = 16, 1000, 200, 512, 511
    rng = np.random.default_rng(0xC0FFEE + 7)
    il = rng.integers(50, T + 1, B).astype(np.int32)
    ll = rng.integers(10, U + 1, B).astype(np.int32)
    il[3], ll[3] = T, U      # one utterance at both maxima
    il[4], ll[4] = 50, 10    # one at both minima
    x, lab, _, _ = _gpu_batch(B, T, U, V, blank, seed=11)
    picks = [3, 4, 9]
    before = x[picks].clone()
    run = _run(x, lab, il, ll, blank)
    for b in range(B):
        assert not run.x[b, il[b]:].any() and not run.x[b, :, ll[b] + 1:].any()
    sub = HipRun.__new__(HipRun)
    sub.costs, sub.x = run.costs[picks], run.x[picks]
    _subset_parity(before, sub, lab[picks], il[picks], ll[picks], blank, [0, 1, 2])


@pytest.mark.parametrize("V,blank", [(64, 63), (62, 39), (1024, 7), (300, 299)])
def test_pruned_rows_are_truly_negligible(V, blank):
    """What the gradient kernels write as exact zero without reading the logits
    again (whole rows whose occupancy is below 2^-28, and all but the blank's
    and the label's entry of rows whose largest probability times the
    occupancy is) must be below 2^-28 in the exact gradient, element by
    element."""
    B, T, U = 1, 300, 80
    x, lab, il, ll = _gpu_batch(B, T, U, V, blank, seed=5)
    xh = x.cpu().numpy()
    run = _run(x, lab, il, ll, blank)
    _, go = orc.rnnt_loss_and_grad(xh, lab.cpu().numpy(), il, ll, blank)
    g = run.x.cpu().numpy()
    zero_rows = ~g.any(-1)
    assert zero_rows.mean() > 0.2                       # the prune does fire at this size
    assert np.abs(go[g == 0]).max() < 2.0 ** -28
    assert np.abs(g - go).max() < 2e-5
    if V % 4 == 0 or V > 128:  # the kernels that also take sparse rows (not the row-per-lane ones of V = 62)
        nnz = np.count_nonzero(g, -1)
        sparse = (nnz >= 1) & (nnz <= 2)
        assert sparse.sum() > 50, sparse.sum()
        # their two entries are the exact ones
        assert np.abs((g - go)[sparse]).max() < 2.0 ** -28


@pytest.mark.parametrize("V,blank", [(64, 63), (256, 0)])
def test_prune_threshold_is_a_runtime_setting_and_is_counted(V, blank):
    """The prune_log2 argument of a call moves the threshold of its gradient pass, rnnt_mi355x_read_stats counts the
    rows that were not read a second time; every element written as zero is bounded by the threshold in the exact
    gradient, and with pruning off the result is the dense one."""
    import ctypes
    from rnnt_amd import _lib
    from tests.hip_runner import _p
    B, T, U = 2, 240, 60
    x, lab, il, ll = _gpu_batch(B, T, U, V, blank, seed=17)
    il[1], ll[1] = 200, 41
    xh = x.cpu().numpy()
    _, go = orc.rnnt_loss_and_grad(xh, lab.cpu().numpy(), il, ll, blank)
    cells_expected = int((il.astype(np.int64) * (ll.astype(np.int64) + 1)).sum())
    L = _lib.lib()
    fracs = {}
    for thr in (float("-inf"), -40.0, -24.0, -16.0):
        run = _run(x.clone(), lab, il, ll, blank, prune_log2=thr)
        cells, pruned = _lib.read_stats(_p(run.ws), run.stream)
        assert cells == cells_expected
        g = run.x.cpu().numpy()
        lattice = np.zeros(g.shape[:3], bool)
        for b in range(B):
            lattice[b, :il[b], :ll[b] + 1] = True
        unread_rows = (np.count_nonzero(g, -1) <= 2) & lattice   # pruned rows, and sparse ones (two exact entries)
        if thr > float("-inf"):  # (with pruning off, far-off rows still underflow to zeros)
            assert unread_rows.sum() == pruned
        fracs[thr] = pruned / cells
        if pruned:
            assert np.abs(go[(g == 0) & lattice[..., None]]).max() <= 2.0 ** thr * 1.001
        assert np.abs(g - go).max() < 2e-5 + 2.0 ** thr
    assert L.rnnt_mi355x_get_prune_log2() == _lib.DEFAULT_PRUNE_LOG2  # per-call thresholds leave the default alone
    assert fracs[float("-inf")] == 0.0
    assert fracs[-40.0] > 0.1 and fracs[-16.0] > fracs[-24.0] > fracs[-40.0]


def test_prune_threshold_is_per_call_also_on_two_streams():
    """Two calls enqueued back to back on two streams with different thresholds (and a third process default in
    force) each get their own result: bit-equal to the same call made alone."""
    from rnnt_amd import _lib
    B, T, U, V, blank = 2, 200, 50, 256, 255
    x, lab, il, ll = _gpu_batch(B, T, U, V, blank, seed=23)
    alone = {thr: _run(x.clone(), lab, il, ll, blank, prune_log2=thr).x.clone() for thr in (float("-inf"), -16.0)}
    assert not torch.equal(alone[float("-inf")], alone[-16.0])
    dev = x.device
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    xa, xb = x.clone(), x.clone()
    torch.cuda.synchronize()
    L = _lib.lib()
    try:
        assert L.rnnt_mi355x_set_prune_log2(-20.0) == 0
        ra = _run(xa, lab, il, ll, blank, prune_log2=float("-inf"), stream=s1)
        rb = _run(xb, lab, il, ll, blank, prune_log2=-16.0, stream=s2)
        rc = _run(x.clone(), lab, il, ll, blank)  # no argument: the process default, -20
        torch.cuda.synchronize()
    finally:
        assert L.rnnt_mi355x_set_prune_log2(_lib.DEFAULT_PRUNE_LOG2) == 0
    assert torch.equal(ra.x, alone[float("-inf")]) and torch.equal(rb.x, alone[-16.0])
    assert torch.equal(rc.x, _run(x.clone(), lab, il, ll, blank, prune_log2=-20.0).x)


@pytest.mark.parametrize("B,T,U,V,blank", [(32, 500, 100, 62, 39), (16, 400, 100, 1024, 1023)])
def test_fused_joint_full_size_equals_materialised_path(B, T, U, V, blank):
    """BASELINE shapes 2 and 3 (a quarter of config 3's batch): the additive-joint path (matrix-core
    contractions, nothing of size T*U*V) against broadcast add + the materialised loss + reductions."""
    import rnnt_amd
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + V)
    f = torch.empty((B, T, V), device=dev).normal_(generator=g)
    gg = torch.empty((B, U + 1, V), device=dev).normal_(generator=g)
    lab = torch.randint(0, V - 1, (B, U), generator=g, device=dev, dtype=torch.int32)
    lab = lab + (lab >= blank).to(torch.int32)
    il = torch.randint(T // 2, T + 1, (B,), generator=g, device=dev, dtype=torch.int32)
    ll = torch.randint(U // 2, U + 1, (B,), generator=g, device=dev, dtype=torch.int32)
    il[0], ll[0] = T, U
    f1, g1 = f.clone().requires_grad_(True), gg.clone().requires_grad_(True)
    c1 = rnnt_amd.rnnt_joint_loss(f1, g1, lab, il, ll, blank)
    c1.mean().backward()
    f2, g2 = f.clone().requires_grad_(True), gg.clone().requires_grad_(True)
    c2 = rnnt_amd.rnnt_loss(f2.unsqueeze(2) + g2.unsqueeze(1), lab, il, ll, blank, inplace=True)
    c2.mean().backward()
    torch.testing.assert_close(c1, c2, rtol=COST_RTOL, atol=0)
    assert (f1.grad - f2.grad).abs().max().item() < GRAD_ATOL
    assert (g1.grad - g2.grad).abs().max().item() < GRAD_ATOL
    # frames / label positions beyond an utterance's lengths get exact zeros
    for b in range(B):
        assert not f1.grad[b, int(il[b]):].any() and not g1.grad[b, int(ll[b]) + 1:].any()
    # exact parity of one utterance with the float64 oracle
    b = 1
    Tb, Ub = int(il[b]), int(ll[b])
    joint = f[b, :Tb, None, :].double().cpu().numpy() + gg[b, None, :Ub + 1, :].double().cpu().numpy()
    co, go = orc.rnnt_loss_and_grad(joint[None], lab[b:b + 1, :].cpu().numpy(), [Tb], [Ub], blank, [1.0 / B])
    np.testing.assert_allclose(c1[b].item(), co[0], rtol=COST_RTOL)
    assert np.abs(f1.grad[b, :Tb].cpu().numpy() - go[0].sum(1)).max() < GRAD_ATOL
    assert np.abs(g1.grad[b, :Ub + 1].cpu().numpy() - go[0].sum(0)).max() < GRAD_ATOL
