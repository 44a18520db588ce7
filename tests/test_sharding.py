"""Host logic of the utterance-sharded data-parallel path, on CPU: the
length-balanced partitioner, per-rank cropping, and the two-scalar all-reduce
over gloo with world_size 2.  Per-shard numbers come from the oracle (the HIP
path needs a GPU; its sharded run is covered in tests/test_gpu_torch_api.py)."""
import os
import socket
import sys

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


def test_default_partition_is_cheap_on_the_host():
    """shard_batch runs on every rank for every batch: its default partition must stay far below a training step
    (the exhaustive "auto" search is opt-in)."""
    import time
    for B, limit in ((128, 0.15), (256, 0.45), (512, 1.5)):
        rng = np.random.default_rng(B)
        il = rng.integers(50, 1001, B).tolist()
        ll = rng.integers(10, 201, B).tolist()
        dt = float("inf")
        for _ in range(3):  # the best of three: the machine that runs this may be busy
            sharding._PARTITION_CACHE.clear()
            t0 = time.perf_counter()
            parts = sharding.partition(il, ll, 8, 512, "bucketed")
            dt = min(dt, time.perf_counter() - t0)
        assert sorted(i for s in parts for i in s) == list(range(B))
        assert dt < limit, f"B={B}: {dt:.3f} s"
    # "auto" is bounded by a count of model evaluations, not by a clock (every rank stops at the same place)
    sharding._PARTITION_CACHE.clear()
    t0 = time.perf_counter()
    p1 = sharding.partition(il, ll, 8, 512, "auto")
    assert time.perf_counter() - t0 < 5.0
    sharding._PARTITION_CACHE.clear()
    assert sharding.partition(il, ll, 8, 512, "auto") == p1


def test_numpy_bucketed_cut_is_the_exhaustive_one():
    """The vectorised dynamic programme against a plain triple loop on small batches."""
    rng = np.random.default_rng(3)
    for n, k in ((7, 3), (12, 4), (5, 8)):
        il = rng.integers(50, 1001, n).tolist()
        ll = rng.integers(0, 201, n).tolist()
        for key in ("T", "U", "cells"):
            p = sharding.partition_bucketed(il, ll, k, 512, key)
            got = max(sharding.shard_seconds(il, ll, s, 512) for s in p)
            order = sorted(range(n), key={"T": lambda i: (-il[i], -ll[i], i), "U": lambda i: (-ll[i], -il[i], i),
                                          "cells": lambda i: (-il[i] * (ll[i] + 1), i)}[key])
            best = {(0, 0): 0.0}
            for s in range(1, min(k, n) + 1):
                for b in range(s, n + 1):
                    best[(s, b)] = min(max(best[(s - 1, a)], sharding.shard_seconds(il, ll, order[a:b], 512))
                                       for a in range(s - 1, b) if (s - 1, a) in best)
            assert abs(got - best[(min(k, n), n)]) <= 1e-12 * got


def test_allgather_costs_two_ranks_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "from rnnt_amd import sharding\n"
        "dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=2)\n"
        "r = dist.get_rank()\n"
        "il = [100, 50, 70, 90, 30]; ll = [10, 5, 7, 9, 3]\n"
        "idx = sharding.partition(il, ll, 2, 62, 'bucketed')[r]\n"
        "local = torch.tensor([float(10 * i + 1) for i in idx])\n"
        "full = sharding.allgather_costs(local, idx, 5)\n"
        "assert full.tolist() == [1.0, 11.0, 21.0, 31.0, 41.0], full\n"
        "dist.destroy_process_group()\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", WORLD_SIZE="2", GLOO_SOCKET_IFNAME="lo")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    assert [p.wait(timeout=120) for p in procs] == [0, 0]
