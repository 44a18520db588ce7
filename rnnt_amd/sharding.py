"""Utterance sharding across the GPUs of one node.

The reference's only parallelism is in-process batch slicing
(``multi_gpu_model``, reference rnnt/keras_model.py:203-206, with ``lr /= N``
and ``batch *= N`` at reference rnnt/base/common.py:177-188).  Utterances are
independent in the transducer loss, so here each rank (one process per GPU)
owns a subset of the batch, runs the three kernels locally and exchanges only
two scalars (loss sum, squared gradient norm) with one all-reduce over RCCL.
No logits, lattice or gradient bytes ever cross xGMI.
"""
from __future__ import annotations

from typing import List, Sequence

import torch


def lattice_cost(input_lengths: Sequence[int], label_lengths: Sequence[int]) -> List[int]:
    """Work of each utterance: its lattice cells T_b * (U_b + 1) (V is common)."""
    return [int(t) * (int(u) + 1) for t, u in zip(input_lengths, label_lengths)]


def partition_lpt(costs: Sequence[int], n_shards: int) -> List[List[int]]:
    """Longest-processing-time-first greedy partition: utterances sorted by
    decreasing cost, each given to the currently lightest shard.  Returns
    n_shards index lists (ascending inside a shard); deterministic, so every
    rank computes the same partition without communicating."""
    if n_shards < 1:
        raise ValueError("n_shards must be >= 1")
    order = sorted(range(len(costs)), key=lambda i: (-int(costs[i]), i))
    loads = [0] * n_shards
    shards: List[List[int]] = [[] for _ in range(n_shards)]
    for i in order:
        k = min(range(n_shards), key=lambda s: (loads[s], len(shards[s]), s))
        shards[k].append(i)
        loads[k] += int(costs[i])
    return [sorted(s) for s in shards]


def shard_batch(acts, labels, input_lengths, label_lengths, rank: int, world_size: int):
    """This rank's slice of a batch, cropped to the shard's own max T and U+1
    (each rank pads only to its local maxima).  Returns (acts, labels, in_len,
    lab_len, indices); acts is a fresh contiguous tensor when cropped."""
    il = [int(v) for v in torch.as_tensor(input_lengths).reshape(-1).tolist()]
    ll = [int(v) for v in torch.as_tensor(label_lengths).reshape(-1).tolist()]
    idx = partition_lpt(lattice_cost(il, ll), world_size)[rank]
    if not idx:
        return None, None, None, None, idx
    sel = torch.as_tensor(idx, dtype=torch.long, device=acts.device if isinstance(acts, torch.Tensor) else None)
    t_max = max(il[i] for i in idx)
    u_max = max(ll[i] for i in idx)
    a = acts.index_select(0, sel)[:, :t_max, : u_max + 1, :].contiguous()
    lab = torch.as_tensor(labels).reshape(len(il), -1)
    lab = lab.index_select(0, sel.to(lab.device))[:, :u_max].contiguous()
    return (a, lab, torch.tensor([il[i] for i in idx], dtype=torch.int32),
            torch.tensor([ll[i] for i in idx], dtype=torch.int32), idx)


def allreduce_scalars(local: torch.Tensor, group=None, async_op: bool = False):
    """Sum a small float64 vector ({loss sum, squared grad norm}) over all
    ranks: the only collective of the data-parallel path (8-16 bytes, so purely
    latency-bound; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU).
    Returns the work handle when async_op, else the reduced tensor."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None if async_op else local
    work = dist.all_reduce(local, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return work if async_op else local
