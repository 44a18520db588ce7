"""Host logic of the utterance-sharded data-parallel path, on CPU: the
length-balanced partitioner, per-rank cropping, and the two-scalar all-reduce
over gloo with world_size 2.  Per-shard numbers come from the oracle (the HIP
path needs a GPU; its sharded run is covered in tests/test_gpu_torch_api.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import rnnt_oracle as orc
from rnnt_amd import sharding
from rnnt_amd.sharding import allreduce_scalars, lattice_cost, partition_lpt, shard_batch


def test_lpt_covers_every_utterance_once_and_balances():
    rng = np.random.default_rng(0)
    T = rng.integers(50, 1001, 128)
    U = rng.integers(10, 201, 128)
    costs = lattice_cost(T, U)
    for n in (1, 2, 4, 8):
        shards = partition_lpt(costs, n)
        assert sorted(i for s in shards for i in s) == list(range(128))
        loads = [sum(costs[i] for i in s) for s in shards]
        assert max(loads) <= sum(costs) / n + max(costs)          # greedy bound
        assert max(loads) / (sum(costs) / n) < 1.02                # config-5-like batches balance within 2 %


def test_lpt_is_deterministic_and_handles_more_shards_than_items():
    assert partition_lpt([5, 3, 8, 1, 9, 2], 3) == partition_lpt([5, 3, 8, 1, 9, 2], 3)
    shards = partition_lpt([7, 7], 4)
    assert sorted(map(len, shards)) == [0, 0, 1, 1]
    with pytest.raises(ValueError):
        partition_lpt([1], 0)


def test_shard_batch_crops_to_local_maxima():
    rng = np.random.default_rng(1)
    B, T, U, V = 6, 12, 5, 7
    acts = torch.from_numpy(rng.standard_normal((B, T, U + 1, V)).astype(np.float32))
    labels = torch.from_numpy(rng.integers(0, V, (B, U)).astype(np.int32))
    il = torch.tensor([12, 3, 7, 12, 5, 9])
    ll = torch.tensor([5, 1, 2, 4, 0, 3])
    seen = []
    for rank in range(2):
        a, lab, i2, l2, idx = shard_batch(acts, labels, il, ll, rank, 2)
        seen += idx
        assert a.shape == (len(idx), int(i2.max()), int(l2.max()) + 1, V) and a.is_contiguous()
        for k, i in enumerate(idx):
            assert torch.equal(a[k, : il[i], : ll[i] + 1], acts[i, : il[i], : ll[i] + 1])
            assert torch.equal(lab[k, : ll[i]], labels[i, : ll[i]])
    assert sorted(seen) == list(range(B))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, x, lab, il, ll, blank, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, lb, i2, l2, idx = shard_batch(torch.from_numpy(x), torch.from_numpy(lab), torch.from_numpy(il), torch.from_numpy(ll), rank, world)
    gs = np.full(len(idx), 1.0 / len(il))
    costs, grad = orc.rnnt_loss_and_grad(a.numpy(), lb.numpy(), i2.numpy(), l2.numpy(), blank, gs)
    local = torch.tensor([costs.sum(), (grad ** 2).sum()], dtype=torch.float64)
    work = allreduce_scalars(local, async_op=True)
    work.wait()
    out_q.put((rank, local.tolist(), idx))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_equals_unsharded():
    rng = np.random.default_rng(2)
    B, T, U, V, blank = 5, 9, 4, 6, 5
    x = rng.standard_normal((B, T, U + 1, V)).astype(np.float32)
    lab = rng.integers(0, V - 1, (B, U)).astype(np.int32)
    il = np.array([9, 4, 7, 9, 2], np.int32)
    ll = np.array([4, 1, 0, 3, 2], np.int32)
    costs, grad = orc.rnnt_loss_and_grad(x, lab, il, ll, blank, np.full(B, 1.0 / B))
    want = [costs.sum(), (grad ** 2).sum()]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, x, lab, il, ll, blank, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, vals, idx in got:
        np.testing.assert_allclose(vals, want, rtol=1e-12)  # every rank holds the global sums
    assert sorted(i for _, _, idx in got for i in idx) == list(range(B))


def test_allreduce_is_identity_without_process_group():
    t = torch.tensor([1.5, 2.5], dtype=torch.float64)
    assert allreduce_scalars(t) is t
    assert allreduce_scalars(t, async_op=True) is None


def _ragged_lengths(seed, B=128):
    rng = np.random.default_rng(seed)
    return rng.integers(50, 1001, B).tolist(), rng.integers(10, 201, B).tolist()


@pytest.mark.parametrize("method", ["cells", "bucketed", "greedy", "auto"])
@pytest.mark.parametrize("n", [1, 2, 3, 8])
def test_every_partitioner_covers_each_utterance_once_and_is_deterministic(method, n):
    il, ll = _ragged_lengths(7, B=50)
    p1 = sharding.partition(il, ll, n, 512, method)
    p2 = sharding.partition(list(il), list(ll), n, 512, method)
    assert p1 == p2
    assert len(p1) == n
    assert sorted(i for s in p1 for i in s) == list(range(50))
    assert all(s == sorted(s) for s in p1)


def test_partitioners_handle_fewer_utterances_than_shards():
    for method in ("cells", "bucketed", "greedy", "auto"):
        p = sharding.partition([100, 50, 70], [10, 5, 7], 8, 512, method)
        assert sorted(i for s in p for i in s) == [0, 1, 2] and len(p) == 8


def test_length_aware_partition_streams_less_padding_than_cell_balanced_lpt():
    """Ragged batch of the B=128, T in [50,1000], U in [10,200] configuration on 8 ranks: balancing lattice cells
    puts long and short utterances on one rank, whose padded tensor (what the gradient pass writes) then is mostly
    padding; the default partition must cut both the padded volume and the modelled slowest rank."""
    il, ll = _ragged_lengths(0xC0FFEE + 7)
    lpt = sharding.partition(il, ll, 8, 512, "cells")
    auto = sharding.partition(il, ll, 8, 512, "auto")
    rows = lambda p: [sharding.padded_rows(il, ll, s) for s in p]
    secs = lambda p: max(sharding.shard_seconds(il, ll, s, 512) for s in p)
    assert sum(rows(auto)) < 0.75 * sum(rows(lpt))
    assert max(rows(auto)) < 0.9 * max(rows(lpt))
    assert secs(auto) < 0.85 * secs(lpt)
    # uniform lengths: nothing to gain, equal counts
    assert [len(s) for s in sharding.partition([400] * 64, [100] * 64, 8, 1024, "auto")] == [8] * 8


def test_refinement_never_makes_the_slowest_shard_slower():
    il, ll = _ragged_lengths(3, B=40)
    base = sharding.partition(il, ll, 4, 512, "cells")
    ref = sharding.refine_partition(il, ll, base, 512)
    worst = lambda p: max(sharding.shard_seconds(il, ll, s, 512) for s in p)
    assert worst(ref) <= worst(base)
    assert sorted(i for s in ref for i in s) == list(range(40))
