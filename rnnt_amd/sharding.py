"""Utterance sharding across the GPUs of one node.

A shard's time is set by what it streams, not by its lattice cells alone: every
rank pads its utterances to its own longest T and U+1 and the gradient pass
writes (zero-fills) all of that, so a partition that mixes long and short
utterances on one rank pays for the padding.  :func:`partition` therefore
chooses between cell-balanced LPT, a length-bucketed cut and a greedy on a
measured time model, by the modelled time of the slowest shard.

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

import numpy as np
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


# Constants of the per-shard time model, measured on one MI355X with the kernels of this library: the first pass
# reads the logits of every lattice cell, the gradient pass writes every row of the shard's padded tensor and
# re-reads the rows it does not prune, the two lattice sweeps run T+U dependent steps whose length depends on how
# many label columns a lane owns.  Only the ratios matter for choosing a partition.
MODEL_READ_BPS = 6.7e12
MODEL_WRITE_BPS = 5.4e12
MODEL_UNPRUNED = 0.5
MODEL_LAUNCH_S = 20e-6


MODEL_SKEW_BPS = 1.5e12  # re-ordering pass in front of the wide sweep (16 bytes per padded lattice position)


def _sweep_step_s(max_u1: int) -> float:
    """Seconds per anti-diagonal of the lattice sweeps: one wavefront per direction up to 128 label positions
    (one or two columns per lane), above that one wavefront per 64 columns (measured 82 ns at 200 positions;
    160 ns at 400, where eight wavefronts' stores share one compute unit)."""
    if max_u1 <= 64:
        return 60e-9
    if max_u1 <= 128:
        return 92e-9
    return 66e-9 + 0.08e-9 * max_u1 + 0.46e-9 * max(max_u1 - 256, 0)


def shard_seconds(input_lengths: Sequence[int], label_lengths: Sequence[int], indices: Sequence[int], V: int) -> float:
    """Modelled time of one forward+backward over the utterances `indices`,
    padded to their own max T and max U+1."""
    if not indices:
        return 0.0
    t_max = max(int(input_lengths[i]) for i in indices)
    u1_max = max(int(label_lengths[i]) for i in indices) + 1
    cells = sum(int(input_lengths[i]) * (int(label_lengths[i]) + 1) for i in indices)
    return _model_seconds(len(indices), t_max, u1_max, cells, V)


def _model_seconds(n: int, t_max: int, u1_max: int, cells: int, V: int) -> float:
    rows = n * t_max * u1_max
    row_bytes = 4.0 * V
    skew = 16.0 * rows / MODEL_SKEW_BPS if u1_max > 128 else 0.0
    return (cells * row_bytes * (1.0 + MODEL_UNPRUNED) / MODEL_READ_BPS + rows * row_bytes / MODEL_WRITE_BPS +
            (t_max + u1_max) * _sweep_step_s(u1_max) + skew + MODEL_LAUNCH_S)


def _model_seconds_np(n, t_max, u1_max, cells, V: int):
    """_model_seconds on arrays (same arithmetic, element by element)."""
    rows = n * t_max * u1_max
    row_bytes = 4.0 * V
    step = np.where(u1_max <= 64, 60e-9, np.where(u1_max <= 128, 92e-9,
                    66e-9 + 0.08e-9 * u1_max + 0.46e-9 * np.maximum(u1_max - 256, 0)))
    skew = np.where(u1_max > 128, 16.0 * rows / MODEL_SKEW_BPS, 0.0)
    return (cells * row_bytes * (1.0 + MODEL_UNPRUNED) / MODEL_READ_BPS + rows * row_bytes / MODEL_WRITE_BPS +
            (t_max + u1_max) * step + skew + MODEL_LAUNCH_S)


def padded_rows(input_lengths: Sequence[int], label_lengths: Sequence[int], indices: Sequence[int]) -> int:
    """Rows (b, t, u) of the shard's tensor once padded to its own maxima."""
    if not indices:
        return 0
    return (len(indices) * max(int(input_lengths[i]) for i in indices) *
            (max(int(label_lengths[i]) for i in indices) + 1))


def partition_bucketed(input_lengths: Sequence[int], label_lengths: Sequence[int], n_shards: int, V: int,
                       sort_by: str = "best") -> List[List[int]]:
    """Length-bucketed partition: utterances sorted by length and cut into
    n_shards consecutive runs so that the largest modelled shard time is as
    small as possible (exact for that order, by dynamic programming over the
    cut points).  Utterances of similar length share a shard, so little of a
    shard's padded tensor is padding.  sort_by: "T" (frames), "U" (labels),
    "cells" (T*(U+1)) or "best" (the best of the three).  Deterministic."""
    if n_shards < 1:
        raise ValueError("n_shards must be >= 1")
    if sort_by == "best":
        best, best_t = None, None
        for key in ("T", "U", "cells"):
            p = partition_bucketed(input_lengths, label_lengths, n_shards, V, key)
            t = max(shard_seconds(input_lengths, label_lengths, s, V) for s in p)
            if best_t is None or t < best_t * (1.0 - 1e-9):
                best, best_t = p, t
        return best
    n = len(input_lengths)
    keys = {"T": lambda i: (-int(input_lengths[i]), -int(label_lengths[i]), i),
            "U": lambda i: (-int(label_lengths[i]), -int(input_lengths[i]), i),
            "cells": lambda i: (-int(input_lengths[i]) * (int(label_lengths[i]) + 1), i)}
    if sort_by not in keys:
        raise ValueError(f"unknown sort key {sort_by!r}")
    order = sorted(range(n), key=keys[sort_by])
    k = min(n_shards, n) if n else 0
    if n == 0:
        return [[] for _ in range(n_shards)]
    # cost[a][b] = modelled time of order[a:b] for a < b (whole arrays at once: the triple loop in Python took
    # seconds for a few hundred utterances, on every rank and every batch)
    t_sorted = np.asarray([int(input_lengths[i]) for i in order], dtype=np.float64)
    u_sorted = np.asarray([int(label_lengths[i]) + 1 for i in order], dtype=np.float64)
    a_idx = np.arange(n)[:, None]
    b_idx = np.arange(n)[None, :]
    live = b_idx >= a_idx                                   # entry (a, j): utterance j of the order belongs to runs from a
    t_run = np.maximum.accumulate(np.where(live, t_sorted[None, :], 0.0), axis=1)
    u_run = np.maximum.accumulate(np.where(live, u_sorted[None, :], 0.0), axis=1)
    c_run = np.cumsum(np.where(live, (t_sorted * u_sorted)[None, :], 0.0), axis=1)
    count = np.where(live, b_idx - a_idx + 1, 0).astype(np.float64)
    inf = float("inf")
    cost = np.full((n + 1, n + 1), inf)
    cost[:n, 1:] = np.where(live, _model_seconds_np(count, t_run, u_run, c_run, V), inf)
    best_prev = np.full(n + 1, inf)
    best_prev[0] = 0.0
    cut = np.zeros((k + 1, n + 1), dtype=np.int64)
    for s in range(1, k + 1):
        m = np.maximum(best_prev[:, None], cost)            # (a, b): slowest shard if the s-th run is order[a:b]
        cut[s] = np.argmin(m, axis=0)                       # (first minimum: the smallest a, as a scan would find)
        best_prev = m[cut[s], np.arange(n + 1)]
    shards: List[List[int]] = [[] for _ in range(n_shards)]
    b = n
    for s in range(k, 0, -1):
        a = int(cut[s][b])
        shards[s - 1] = sorted(order[a:b])
        b = a
    return shards


def partition_greedy_model(input_lengths: Sequence[int], label_lengths: Sequence[int], n_shards: int, V: int) -> List[List[int]]:
    """Longest-first greedy on the modelled time: utterances by decreasing T,
    each given to the shard whose modelled time after taking it is smallest."""
    if n_shards < 1:
        raise ValueError("n_shards must be >= 1")
    order = sorted(range(len(input_lengths)), key=lambda i: (-int(input_lengths[i]), -int(label_lengths[i]), i))
    shards: List[List[int]] = [[] for _ in range(n_shards)]
    state = [[0, 0, 0, 0] for _ in range(n_shards)]  # per shard: utterances, max T, max U+1, cells
    for i in order:
        t, u1 = int(input_lengths[i]), int(label_lengths[i]) + 1

        def after(s):
            c, tm, um, ce = state[s]
            return _model_seconds(c + 1, max(tm, t), max(um, u1), ce + t * u1, V)

        k = min(range(n_shards), key=lambda s: (after(s), len(shards[s]), s))
        shards[k].append(i)
        c, tm, um, ce = state[k]
        state[k] = [c + 1, max(tm, t), max(um, u1), ce + t * u1]
    return [sorted(s) for s in shards]


def refine_partition(input_lengths: Sequence[int], label_lengths: Sequence[int], shards: List[List[int]], V: int,
                     max_rounds: int = 200, max_evals: int = 4000) -> List[List[int]]:
    """Local search on a partition: move one utterance out of the slowest shard,
    or swap it with one of another shard, whenever that lowers the modelled time
    of the slowest shard (first improvement in a fixed scan order, so the result
    is deterministic).  The search stops after max_rounds improvements or
    max_evals evaluations of the time model -- a count, not a clock: every rank
    must stop at the same place."""
    shards = [list(s) for s in shards]
    n = len(shards)
    evals = [0]

    class _Spent(Exception):
        pass

    def sec(s):
        evals[0] += 1
        if evals[0] > max_evals:
            raise _Spent
        return shard_seconds(input_lengths, label_lengths, s, V)

    try:
        return _refine(shards, n, sec, max_rounds)
    except _Spent:
        return [sorted(s) for s in shards]  # (the lists are only ever replaced together, after both evaluations)


def _refine(shards, n, sec, max_rounds):

    times = [sec(s) for s in shards]
    for _ in range(max_rounds):
        worst = max(range(n), key=lambda k: (times[k], -k))
        t_worst = times[worst]
        done = False
        for i in sorted(shards[worst]):
            rest = [j for j in shards[worst] if j != i]
            t_rest = sec(rest)
            for k in range(n):
                if k == worst:
                    continue
                # move i -> k
                t_k = sec(shards[k] + [i])
                if max(t_rest, t_k) < t_worst * (1.0 - 1e-6):
                    shards[worst], shards[k] = rest, shards[k] + [i]
                    times[worst], times[k] = t_rest, t_k
                    done = True
                    break
                # swap i <-> j
                for j in sorted(shards[k]):
                    a = rest + [j]
                    b = [m for m in shards[k] if m != j] + [i]
                    ta, tb = sec(a), sec(b)
                    if max(ta, tb) < t_worst * (1.0 - 1e-6):
                        shards[worst], shards[k] = a, b
                        times[worst], times[k] = ta, tb
                        done = True
                        break
                if done:
                    break
            if done:
                break
        if not done:
            break
    return [sorted(s) for s in shards]


def partition(input_lengths: Sequence[int], label_lengths: Sequence[int], n_shards: int, V: int = 1024,
              method: str = "auto") -> List[List[int]]:
    """The partition every rank derives for itself (no communication):
    "cells" = LPT on lattice cells, "bucketed" = partition_bucketed, "greedy" =
    partition_greedy_model, "auto" = whichever of the three has the smallest
    modelled slowest shard (ties: in that order), then improved by
    refine_partition."""
    il = [int(v) for v in input_lengths]
    ll = [int(v) for v in label_lengths]
    key = (tuple(il), tuple(ll), int(n_shards), int(V), method)
    hit = _PARTITION_CACHE.get(key)
    if hit is not None:
        return [list(s) for s in hit]
    out = _partition(il, ll, n_shards, V, method)
    if len(_PARTITION_CACHE) >= 16:
        _PARTITION_CACHE.pop(next(iter(_PARTITION_CACHE)))
    _PARTITION_CACHE[key] = [list(s) for s in out]
    return out


_PARTITION_CACHE: dict = {}  # the last few batches' partitions (validation sets and fixed-length batches repeat)


def _partition(il, ll, n_shards, V, method):
    cands = {
        "cells": lambda: partition_lpt(lattice_cost(il, ll), n_shards),
        "bucketed": lambda: partition_bucketed(il, ll, n_shards, V),
        "greedy": lambda: partition_greedy_model(il, ll, n_shards, V),
    }
    if method != "auto":
        if method not in cands:
            raise ValueError(f"unknown partition method {method!r}")
        return cands[method]()
    best, best_t = None, None
    for name in ("cells", "bucketed", "greedy"):
        p = cands[name]()
        t = max(shard_seconds(il, ll, s, V) for s in p)
        if best_t is None or t < best_t * (1.0 - 1e-9):
            best, best_t = p, t
    return refine_partition(il, ll, best, V)


def shard_batch(acts, labels, input_lengths, label_lengths, rank: int, world_size: int, method: str = "bucketed"):
    """This rank's slice of a batch, cropped to the shard's own max T and U+1
    (each rank pads only to its local maxima).  Returns (acts, labels, in_len,
    lab_len, indices); acts is a fresh contiguous tensor when cropped.

    The partition is computed on the host for every batch, on every rank: the
    default is the length-bucketed cut (a few milliseconds for a hundred
    utterances, ~0.15 s for 512; on a ragged batch within 1 % of "auto" by the
    measured slowest rank).  "auto" adds the greedy candidate and a bounded
    local search and costs tens of milliseconds to half a second; "cells" is
    the cheapest (sub-millisecond) and pays for padding on ragged batches."""
    il = [int(v) for v in torch.as_tensor(input_lengths).reshape(-1).tolist()]
    ll = [int(v) for v in torch.as_tensor(label_lengths).reshape(-1).tolist()]
    V = int(acts.shape[-1]) if hasattr(acts, "shape") else 1024
    idx = partition(il, ll, world_size, V, method)[rank]
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


def allgather_costs(local_costs: torch.Tensor, indices: Sequence[int], batch: int, group=None) -> torch.Tensor:
    """Per-utterance costs of the whole batch on every rank, in the batch's own
    order: each rank contributes the costs of its shard (`indices` = the shard's
    utterances, as partition returned them).  One all-reduce of `batch` floats
    into a zero vector -- every slot has exactly one owner, so the sum is a
    gather that needs no second exchange for the index lists (every rank could
    derive them, but a sum needs none).  Optional: the training step itself
    needs only allreduce_scalars."""
    import torch.distributed as dist

    out = torch.zeros(batch, dtype=local_costs.dtype, device=local_costs.device)
    if len(indices):
        out[torch.as_tensor(list(indices), dtype=torch.long, device=out.device)] = local_costs.reshape(-1)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return out


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
