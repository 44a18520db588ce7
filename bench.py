#!/usr/bin/env python3
"""Benchmark of the RNN-T loss hot path (fused forward + in-place backward).

One "step" = one rnnt_mi355x_loss_grad (fused forward + in-place backward; its
_events form, so that the gradient kernels are bracketed by events on the
launch stream) over one batch of synthetic logits already resident in HBM,
followed by the reduction of the two data-parallel scalars and, for N > 1,
their all-reduce over RCCL.

Metric: lattice cells per second, whole job (all ranks), where a cell is one
(b, t, u) with t < T_b and u <= U_b.  Weak scaling: every rank processes its own
full copy of the workload (the batch is sharded by utterance, there is no
data-path collective), so per-GPU work is fixed as N grows.

The logits are overwritten by their gradient in every step, so the steps cycle
through a ring of pre-filled buffers; when the ring is shorter than
warmup+steps the timed region is split into segments, each bracketed by
barrier + synchronize, and the refill between segments is not timed.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak

# BASELINE.json configs (index = config id).  lengths "full" or "ragged".
CONFIGS = {
    1: dict(B=8, T=150, U=40, V=62, blank=39, ragged=False, name="TIMIT graves13 3L-250H shape B=8 T=150 U=40 V=62"),
    2: dict(B=32, T=500, U=100, V=62, blank=39, ragged=False, name="TIMIT-shaped B=32 T=500 U=100 V=62"),
    3: dict(B=64, T=400, U=100, V=1024, blank=1023, ragged=False, name="char/wordpiece B=64 T=400 U=100 V=1024"),
    4: dict(B=32, T=250, U=150, V=4096, blank=4095, ragged=False, name="large-vocab B=32 T=250 U=150 V=4096"),
    5: dict(B=128, T=1000, U=200, V=512, blank=511, ragged=True, name="ragged B=128 T in [50,1000] U in [10,200] V=512"),
}


def bytes_per_cell(V: int) -> int:
    """Algorithmic HBM bytes per lattice cell of the whole path: read the V
    logits once, write the V gradients once in place, plus alpha, beta and the
    two log-probabilities."""
    return 8 * V + 16


def make_lengths(cfg, rank, seed):
    B = cfg["B"]
    if not cfg["ragged"]:
        return np.full(B, cfg["T"], np.int32), np.full(B, cfg["U"], np.int32)
    rng = np.random.default_rng(seed + 2)
    il = rng.integers(50, cfg["T"] + 1, size=B).astype(np.int32)
    ll = rng.integers(10, cfg["U"] + 1, size=B).astype(np.int32)
    return il, ll


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def cpu_baseline(master, labels, il, ll, cfg, max_utts=16, reps=3):
    """Times the fp32 C restatement (oracle/rnnt_cpu.c, kind "port") on the
    host cores over a bounded sample of the same workload."""
    from oracle import cpu_port
    cpu_port.build()
    n = min(max_utts, cfg["B"])
    threads = min(cpu_port.max_threads(), n, 16)
    x = master[:n].cpu().numpy()
    lab = labels[:n].cpu().numpy()
    ilh, llh = il[:n].copy(), ll[:n].copy()
    cells = int((ilh.astype(np.int64) * (llh.astype(np.int64) + 1)).sum())
    best = None
    for _ in range(reps):
        buf = x.copy()
        t0 = time.perf_counter()
        cpu_port.loss_grad(buf, lab, ilh, llh, cfg["blank"], inplace=True, num_threads=threads)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"value": cells / best, "unit": "cells/s", "cores": threads, "kind": "port",
            "sample": f"first {n} utterances of the workload ({cells} cells), best of {reps} runs, "
                      f"fp32 C restatement with OpenMP over utterances"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config id (1..5); 3 is the graded point")
    ap.add_argument("--ring", type=int, default=0, help="logits buffers to cycle through (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=0, help="override B (debug)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    args = ap.parse_args()

    t_start = time.perf_counter()

    def dbg(msg):
        if os.environ.get("BENCH_DEBUG"):
            print(f"[bench rank {os.environ.get('RANK', '0')} +{time.perf_counter() - t_start:6.1f}s] {msg}", file=sys.stderr, flush=True)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    local_rank %= torch.cuda.device_count()  # rehearsals with more ranks than devices share a device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":  # single-node rehearsal: the host name may not resolve
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group(args.backend, rank=rank, world_size=world, **({"device_id": dev} if args.backend == "nccl" else {}))

    dbg("process group ready")
    from rnnt_amd import _lib
    L = _lib.lib()

    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["B"] = args.batch
    B, T, U, V, blank = cfg["B"], cfg["T"], cfg["U"], cfg["V"], cfg["blank"]
    U1 = U + 1
    seed = 0xC0FFEE + args.config + 1000 * rank
    il_h, ll_h = make_lengths(cfg, rank, seed)
    cells = int((il_h.astype(np.int64) * (ll_h.astype(np.int64) + 1)).sum())

    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    master = torch.empty((B, T, U1, V), dtype=torch.float32, device=dev)
    master.normal_(generator=gen)
    labels = torch.randint(0, V - 1, (B, max(U, 1)), generator=gen, device=dev, dtype=torch.int32)
    labels = labels + (labels >= blank).to(torch.int32)
    il = torch.from_numpy(il_h).to(dev)
    ll = torch.from_numpy(ll_h).to(dev)
    grad_scale = torch.full((B,), 1.0 / B, dtype=torch.float32, device=dev)  # Keras batch mean
    costs = torch.empty(B, dtype=torch.float32, device=dev)
    scal = torch.zeros(2, dtype=torch.float64, device=dev)
    ws = torch.empty(_lib.workspace_bytes(B, T, U1), dtype=torch.uint8, device=dev)

    nbytes = master.numel() * 4
    free, _total = torch.cuda.mem_get_info(dev)
    total_steps = args.steps + args.warmup
    ring_n = args.ring if args.ring > 0 else max(1, min(total_steps, int((free - (8 << 30)) // nbytes), 24))
    ring = [torch.empty_like(master) for _ in range(ring_n)]

    stream = torch.cuda.current_stream(dev)
    sp = ctypes.c_void_p(stream.cuda_stream)
    meta = (B, T, U1, V, blank)

    def refill():
        for r in ring:
            r.copy_(master)

    def d2d_copy_rate():
        """Device-to-device copy bandwidth of this box (read + write bytes per
        second), the secondary denominator next to the spec peak."""
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ring[0].copy_(master)
        a.record(stream)
        ring[0].copy_(master)
        b.record(stream)
        b.synchronize()
        return 2 * nbytes / (a.elapsed_time(b) * 1e-3) / 1e9

    ev_pairs = []
    ev_steps = []  # (begin, end) of every timed step on the launch stream

    def new_event():
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)  # materialises the hipEvent_t; the library re-records it
        return e

    def step(buf, timed):
        e0 = e1 = s0 = None
        if timed:
            e0, e1 = new_event(), new_event()
            s0 = new_event()
        rc = L.rnnt_mi355x_loss_grad_events(_p(buf), _p(labels), _p(il), _p(ll), *meta, _p(grad_scale), _p(costs),
                                            _p(ws), sp, ctypes.c_void_p(e0.cuda_event) if timed else None,
                                            ctypes.c_void_p(e1.cuda_event) if timed else None)
        if timed:
            ev_pairs.append((e0, e1))
        rc |= L.rnnt_mi355x_reduce_scalars(_p(costs), B, _p(ws), _p(scal), sp)
        if timed:
            ev_steps.append((s0, new_event()))
        if rc != 0:
            raise RuntimeError(f"rnnt_mi355x call failed: status {rc}")
        if world > 1:
            if args.backend == "nccl":
                dist.all_reduce(scal)  # RCCL, enqueued behind the step on the device
            else:  # gloo rehearsal: reduce a host copy
                host = scal.cpu()
                dist.all_reduce(host)

    # barrier = a 1-element all-reduce (on the device for RCCL, on the host for gloo rehearsals)
    bar = torch.zeros(1, dtype=torch.float32, device=dev if args.backend == "nccl" else "cpu")

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.all_reduce(bar)
            torch.cuda.synchronize(dev)

    dbg(f"buffers ready (ring {ring_n})")
    # warmup
    done = 0
    while done < args.warmup:
        refill()
        n = min(ring_n, args.warmup - done)
        for i in range(n):
            step(ring[i], False)
        done += n
    dbg("warmup done")
    # timed: segments of at most ring_n steps, refill between segments untimed
    elapsed = 0.0
    done = 0
    while done < args.steps:
        refill()
        n = min(ring_n, args.steps - done)
        fence()
        t0 = time.perf_counter()
        for i in range(n):
            step(ring[i], True)
        fence()
        elapsed += time.perf_counter() - t0
        done += n
    dbg("timed region done")
    t_max = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed = float(t_max.item())
    cells_all = torch.tensor([cells], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(cells_all)
    total_cells = float(cells_all.item())

    flags = ctypes.c_int32(0)
    rc = L.rnnt_mi355x_read_status(_p(ws), ctypes.byref(flags), sp)
    if rc != 0:
        raise RuntimeError(f"device-side validation failed: flags {flags.value}")
    if not bool(torch.isfinite(costs).all()):
        raise RuntimeError("non-finite costs")

    ms_per_step = elapsed / args.steps * 1e3
    value = total_cells * args.steps / elapsed
    bpc = bytes_per_cell(V)
    grad_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_pairs]))
    step_ms = np.array([a.elapsed_time(b) for a, b in ev_steps])
    d2d = d2d_copy_rate()
    achieved = cells * bpc / (grad_ms * 1e-3) / 1e9

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            ent = tj.get(f"config{args.config}", {})
            if ent.get("B") == B:
                traffic = ent.get("k_grad_hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "RNN-T loss fwd+bwd lattice cells/sec",
        "value": value,
        "unit": "cells/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"config {args.config}: {cfg['name']}{' (B overridden to %d)' % B if args.batch else ''}, "
                        f"fp32 logits ~ N(0,1), blank={blank}, "
                        f"grad_scale=1/B, per-GPU batch (weak scaling: global batch = {B}*n_gpus)",
            "cells_per_gpu": cells, "logits_bytes_per_gpu": nbytes, "ring_buffers": ring_n,
            "parallelism": f"utterance-sharded dp{world}, one 16-byte all-reduce per step" if world > 1 else "single GPU",
        },
        "step_hbm_frac_of_peak": value / world * bpc / (HBM_PEAK_GBS * 1e9),
        "step_ms_device": {"median": float(np.median(step_ms)), "min": float(step_ms.min()), "max": float(step_ms.max())},
        "d2d_copy_GBs": d2d,
        "step_frac_of_d2d_copy": value / world * bpc / (d2d * 1e9),
        "roofline": {
            "bound": "hbm", "kernel": "k_grad (in-place gradient pass)", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes_per_cell": bpc, "kernel_ms": grad_ms,
        },
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(master, labels, il_h, ll_h, cfg)
        out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
