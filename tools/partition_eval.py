#!/usr/bin/env python3
"""Evaluates utterance partitions on ONE GPU ("fake world"): for a BASELINE
configuration split over N ranks, every rank's shard (padded to its own longest
T and U+1, as a rank would hold it) is timed one after another on the same
device.  Reports per method the slowest rank (what the job's step takes), the
spread, the padded volume and, against the unsharded batch on one GPU, the
speed-up N ranks would give.  usage: partition_eval.py --config 5 --world 8"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from rnnt_amd import _lib, sharding  # noqa: E402


def time_shard(L, dev, il_h, ll_h, V, blank, scale_b, reps):
    B, T, U1 = len(il_h), int(il_h.max()), int(ll_h.max()) + 1
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + B * 7 + T)
    master = torch.empty((B, T, U1, V), dtype=torch.float32, device=dev).normal_(generator=gen)
    labels = torch.randint(0, V - 1, (B, max(U1 - 1, 1)), generator=gen, device=dev, dtype=torch.int32)
    labels = labels + (labels >= blank).to(torch.int32)
    il, ll = torch.from_numpy(il_h).to(dev), torch.from_numpy(ll_h).to(dev)
    gs = torch.full((B,), scale_b, dtype=torch.float32, device=dev)
    costs = torch.empty(B, dtype=torch.float32, device=dev)
    scal = torch.zeros(2, dtype=torch.float64, device=dev)
    ws = torch.empty(_lib.workspace_bytes(B, T, U1), dtype=torch.uint8, device=dev)
    buf = torch.empty_like(master)
    sp = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ms = []
    for _ in range(reps + 1):
        buf.copy_(master)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = L.rnnt_mi355x_loss_grad(p(buf), p(labels), p(il), p(ll), B, T, U1, V, blank, p(gs), p(costs), p(ws), sp, float("nan"))
        rc |= L.rnnt_mi355x_reduce_scalars(p(costs), B, p(ws), p(scal), sp)
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0
        ms.append(e0.elapsed_time(e1))
    del master, buf
    return float(np.median(ms[1:])), B * T * U1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=5)
    ap.add_argument("--world", type=int, nargs="+", default=[8])
    ap.add_argument("--methods", default="cells,bucketed,greedy,auto")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.lib()
    cfg = bench.CONFIGS[a.config]
    V, blank = cfg["V"], cfg["blank"]
    il, ll = bench.make_lengths(cfg, 0xC0FFEE + a.config)
    cells = int((il.astype(np.int64) * (ll + 1)).sum())
    t1, rows1 = time_shard(L, dev, il, ll, V, blank, 1.0 / cfg["B"], a.reps)
    print(json.dumps({"config": a.config, "world": 1, "ms": t1, "cells": cells, "padded_GB": rows1 * 4 * V / 1e9,
                      "Mcells_per_s": cells / t1 / 1e3}), flush=True)
    for n in a.world:
        for m in a.methods.split(","):
            parts = sharding.partition(il.tolist(), ll.tolist(), n, V, m)
            ms, gb, model = [], [], []
            for idx in parts:
                if not idx:
                    ms.append(0.0); gb.append(0.0); model.append(0.0)
                    continue
                t, rows = time_shard(L, dev, il[idx], ll[idx], V, blank, 1.0 / cfg["B"], a.reps)
                ms.append(t)
                gb.append(rows * 4 * V / 1e9)
                model.append(sharding.shard_seconds(il, ll, idx, V) * 1e3)
            print(json.dumps({"config": a.config, "world": n, "method": m, "max_rank_ms": max(ms), "min_rank_ms": min(ms),
                              "rank_ms": [round(x, 3) for x in ms], "model_ms": [round(x, 3) for x in model],
                              "padded_GB_max": max(gb), "padded_GB_total": sum(gb), "utterances": [len(s) for s in parts],
                              "speedup_vs_1gpu": t1 / max(ms), "job_Mcells_per_s": cells / max(ms) / 1e3}), flush=True)
            if len(set(il.tolist())) == 1 and len(set(ll.tolist())) == 1:
                break  # uniform lengths: every method gives the same partition


if __name__ == "__main__":
    main()
