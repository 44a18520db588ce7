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
