#!/usr/bin/env python3
"""Benchmark of the RNN-T loss hot path (fused forward + in-place backward).

One "step" = one rnnt_mi355x_loss_grad (its _events form: four events on the
launch stream separate the log-sum-exp pass, the lattice sweep and the gradient
pass) over one batch of synthetic logits already resident in HBM, followed by
the reduction of the two data-parallel scalars and, for N > 1, their all-reduce.

Metric: lattice cells per second, whole job (all ranks), where a cell is one
(b, t, u) with t < T_b and u <= U_b.

Multi-GPU (--gpus N): one process per GPU.  Launched by torchrun the script
reads RANK / LOCAL_RANK / WORLD_SIZE; launched bare (`python bench.py --gpus N`)
it starts the N rank processes itself, before it has touched the GPU.  Default
for N > 1 is the sharded form of the configuration ("strong" scaling): ONE
global batch (the configuration's B, one seed on every rank), every rank derives
the same length-balanced partition, builds only its own shard (padded to the
shard's own longest T and U+1) and exchanges two float64 scalars per step.
`--scaling weak` gives every rank its own full batch instead.

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
import socket
import subprocess
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


def kernel_bytes_per_cell(V: int) -> dict:
    """The same bytes split over the three kernels (the shares add up to
    bytes_per_cell): the first pass reads the row and writes the two
    log-probabilities, the sweep writes alpha and beta, the gradient pass writes
    the row.  The gradient pass's second read of the rows it does not prune is
    traffic of this design, not algorithmic work, and is reported apart."""
    return {"k_lse": 4 * V + 8, "k_sweep": 8, "k_grad": 4 * V}


def make_lengths(cfg, seed):
    """Lengths of the global batch (identical on every rank)."""
    B = cfg["B"]
    if not cfg["ragged"]:
        return np.full(B, cfg["T"], np.int32), np.full(B, cfg["U"], np.int32)
    rng = np.random.default_rng(seed + 2)
    il = rng.integers(50, cfg["T"] + 1, size=B).astype(np.int32)
    ll = rng.integers(10, cfg["U"] + 1, size=B).astype(np.int32)
    return il, ll


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def cpu_baseline(master, labels, il, ll, cfg, reps=3, budget_s=7.0):
    """Times the fp32 C restatement (oracle/rnnt_cpu.c, kind "port") on the
    host cores: the whole batch when that takes a few seconds, else its first
    utterances; median of `reps` runs."""
    from oracle import cpu_port
    cpu_port.build()
    nproc = os.cpu_count() or 1
    threads = int(os.environ.get("OMP_NUM_THREADS") or 0) or cpu_port.max_threads()
    threads = max(1, min(threads, nproc, cfg["B"]))
    cells_b = il.astype(np.int64) * (ll.astype(np.int64) + 1)
    # ~0.14 Mcells/s per thread at V = 1024 on the boxes seen so far; only used to bound the sample
    cap = budget_s * threads * 0.14e6 * 1024.0 / cfg["V"]
    n = int(np.searchsorted(np.cumsum(cells_b), cap, side="right"))
    n = max(min(threads, len(cells_b)), min(n, len(cells_b)))
    threads = min(threads, n)
    x = master[:n].cpu().numpy()
    lab = labels[:n].cpu().numpy()
    ilh, llh = il[:n].copy(), ll[:n].copy()
    cells = int(cells_b[:n].sum())
    buf = np.empty_like(x)
    times = []
    for _ in range(reps):
        np.copyto(buf, x)
        t0 = time.perf_counter()
        cpu_port.loss_grad(buf, lab, ilh, llh, cfg["blank"], inplace=True, num_threads=threads)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    whole = n == len(cells_b)
    return {"value": cells / med, "unit": "cells/s", "cores": threads, "nproc": nproc, "kind": "port",
            "seconds": {"median": med, "min": float(min(times)), "max": float(max(times))},
            "sample": (("the whole batch" if whole else f"first {n} utterances of the batch") +
                       f" ({cells} cells), median of {reps} runs, fp32 C restatement with OpenMP over utterances, "
                       f"{threads} threads on a host reporting {nproc} CPUs")}


# ---------------------------------------------------------------------------
# starting the ranks
# ---------------------------------------------------------------------------

def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n_ranks: int, argv) -> int:
    """Starts n_ranks copies of this script (RANK / LOCAL_RANK / WORLD_SIZE set),
    relays rank 0's output and returns non-zero if any rank failed.  Must run
    before this process has initialised the GPU: a process that has must not
    start GPU children that way (and never re-execs itself)."""
    if torch.cuda.is_initialized():
        raise RuntimeError("bench.py: refusing to start rank processes from a process that has already initialised "
                           "the GPU")
    port = _free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    failed = None
    pending = set(range(n_ranks))
    while pending and failed is None:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0:
                failed = (r, rc)
                break
        if pending and failed is None:
            time.sleep(0.05)
    if failed is not None:  # the others would wait for the dead rank in a collective for ever
        for r in pending:
            procs[r].terminate()
        for r in pending:
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    out = procs[0].stdout.read().decode() if procs[0].stdout else ""
    sys.stdout.write(out)
    sys.stdout.flush()
    if failed is not None:
        print(f"bench.py: rank {failed[0]} exited with status {failed[1]}", file=sys.stderr)
        return 1
    return 0


def counter_bytes_for(config, B, prune_log2, path=None):
    """HBM counter bytes of profiles/traffic.json for this workload -- only if they were measured on the kernels
    that are built now (same kernel sources, batch and threshold); {} otherwise, so that a kernel edit without a
    new profile shows as `traffic: null` instead of an older build's number."""
    path = path or os.path.join(ROOT, "profiles", "traffic.json")
    try:
        from rnnt_amd import build as _build
        ent = json.load(open(path)).get(f"config{config}", {})
        if (ent.get("B") == B and ent.get("prune_log2") == prune_log2 and
                ent.get("csrc_sha16") == _build.sources_digest()):
            return ent
    except Exception:
        pass
    return {}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config id (1..5); 3 is the graded point")
    ap.add_argument("--ring", type=int, default=0, help="logits buffers to cycle through (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the extra legs (pruning off, peaked logits)")
    ap.add_argument("--leg-steps", type=int, default=6)
    ap.add_argument("--batch", type=int, default=0, help="override the global B (debug)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong: the configuration's one global batch sharded over the ranks; weak: a full batch per rank")
    ap.add_argument("--partition", default="auto", help="cells | bucketed | greedy | auto (rnnt_amd.sharding.partition)")
    ap.add_argument("--prune-log2", type=float, default=None, help="pruning threshold of the gradient pass (library default -28)")
    ap.add_argument("--logit-scale", type=float, default=1.0, help="logits ~ N(0,1) * scale")
    ap.add_argument("--graph", choices=("auto", "on", "off"), default="auto", nargs="?", const="on",
                    help="replay each step from a captured HIP graph (kernel times then come from a separate instrumented "
                         "pass).  auto: on for multi-rank runs, whose per-rank steps are short and launch-bound, and for "
                         "workloads of less than 1 GB of logits")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------

def run_rank(args) -> None:
    t_start = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    def dbg(msg):
        if os.environ.get("BENCH_DEBUG"):
            print(f"[bench rank {rank} +{time.perf_counter() - t_start:6.1f}s] {msg}", file=sys.stderr, flush=True)

    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    force_pg = bool(os.environ.get("BENCH_FORCE_PROCESS_GROUP"))  # rehearsal of the N > 1 code path with one rank
    graph_auto = args.graph == "auto"
    args.graph = args.graph == "on" or (graph_auto and (world > 1 or force_pg))
    n_dev = torch.cuda.device_count()  # does not initialise the GPU
    if n_dev < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if world > n_dev and args.backend == "nccl":
        raise SystemExit(f"bench.py: {world} ranks but {n_dev} GPU(s): RCCL needs one GPU per rank "
                         "(rehearse with --backend gloo, ranks then share the device)")
    local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    if world > 1 or force_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "gloo":  # single-node rehearsal: the host name may not resolve
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group(args.backend, rank=rank, world_size=world,
                                **({"device_id": dev} if args.backend == "nccl" else {}))
        world = dist.get_world_size()  # the ranks that actually joined
    multi = world > 1 or force_pg
    on_device_comm = args.backend == "nccl"
    comm_dev = dev if on_device_comm else torch.device("cpu")
    dbg("process group ready")

    from rnnt_amd import _lib, sharding
    L = _lib.lib()
    # the threshold is an argument of every call (the library's process default when none is given here)
    prune_log2 = float(L.rnnt_mi355x_get_prune_log2()) if args.prune_log2 is None else float(args.prune_log2)
    call_prune = [_lib.prune_arg(prune_log2)]  # what launch() passes; the extra legs change it

    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["B"] = args.batch
    V, blank = cfg["V"], cfg["blank"]
    sharded = args.scaling == "strong"
    seed = 0xC0FFEE + args.config + (0 if sharded else 1000 * rank)
    il_g, ll_g = make_lengths(cfg, seed)
    if sharded and world > 1:
        parts = sharding.partition(il_g.tolist(), ll_g.tolist(), world, V, args.partition)
        idx = parts[rank]
        if not idx:
            raise SystemExit(f"bench.py: rank {rank} got no utterance (B={cfg['B']} over {world} ranks)")
    else:
        idx = list(range(cfg["B"]))
    il_h = il_g[idx].copy()
    ll_h = ll_g[idx].copy()
    B = len(idx)
    T = int(il_h.max())   # each rank pads to its own maxima
    U = int(ll_h.max())
    U1 = U + 1
    cells = int((il_h.astype(np.int64) * (ll_h.astype(np.int64) + 1)).sum())
    padded_rows = B * T * U1

    # every utterance has its own generator seed, so a rank can build just its shard
    gen = torch.Generator(device=dev)
    master = torch.empty((B, T, U1, V), dtype=torch.float32, device=dev)
    labels = torch.empty((B, max(U, 1)), dtype=torch.int32, device=dev)
    for j, b in enumerate(idx):
        gen.manual_seed(seed * 65537 + int(b))
        master[j].normal_(generator=gen)
        lab = torch.randint(0, V - 1, (max(U, 1),), generator=gen, device=dev, dtype=torch.int32)
        labels[j] = lab + (lab >= blank).to(torch.int32)
    if args.logit_scale != 1.0:
        master.mul_(args.logit_scale)
    il = torch.from_numpy(il_h).to(dev)
    ll = torch.from_numpy(ll_h).to(dev)
    # Keras batch mean over the global batch
    grad_scale = torch.full((B,), 1.0 / (cfg["B"] if sharded else B), dtype=torch.float32, device=dev)
    costs = torch.empty(B, dtype=torch.float32, device=dev)
    ws = torch.empty(_lib.workspace_bytes(B, T, U1), dtype=torch.uint8, device=dev)

    nbytes = master.numel() * 4
    if graph_auto and nbytes < (1 << 30):  # steps of less than half a millisecond: five launches are 4-16 % of them
        args.graph = True
    free, _total = torch.cuda.mem_get_info(dev)
    total_steps = args.steps + args.warmup
    # ranks that share a device (gloo rehearsals on a one-GPU box) share its memory: each plans with its share
    share = free // max(1, -(-world // n_dev))
    ring_n = args.ring if args.ring > 0 else max(1, min(total_steps, int((share - (8 << 30)) // nbytes), 24))
    if multi:  # every rank must cut its steps into the same segments: the fences between them are collectives
        rn = torch.tensor([ring_n], dtype=torch.int64, device=comm_dev)
        dist.all_reduce(rn, op=dist.ReduceOp.MIN)
        ring_n = int(rn.item())
    ring = [torch.empty_like(master) for _ in range(ring_n)]
    scal = torch.zeros((max(ring_n, 1), 2), dtype=torch.float64, device=dev)  # one slot per step of a segment

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

    def new_event():
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)  # materialises the hipEvent_t; the library re-records it
        return e

    ev_grad = []     # per timed step: (before, after) the gradient pass, the dominant kernel
    ev_kernels = []  # per instrumented step: (before lse, after lse, after sweep, after grad, after scalars)
    pending = []     # all-reduce handles of the current segment

    def launch(i, events=None, on=None):
        """The library calls of one step on buffer i (stream `on`, default the bench's stream)."""
        s = sp if on is None else ctypes.c_void_p(on.cuda_stream)
        arr = None
        if events is not None:
            arr = (ctypes.c_void_p * 4)(*[None if e is None else ctypes.c_void_p(e.cuda_event) for e in events[:4]])
        rc = L.rnnt_mi355x_loss_grad_events(_p(ring[i]), _p(labels), _p(il), _p(ll), *meta, _p(grad_scale), _p(costs),
                                            _p(ws), s, call_prune[0], arr)
        rc |= L.rnnt_mi355x_reduce_scalars(_p(costs), B, _p(ws), _p(scal[i]), s)
        if rc != 0:
            raise RuntimeError(f"rnnt_mi355x call failed: status {rc}")

    graphs = {}

    def step(i, timed):
        """timed: False = no events; True = two events around the gradient pass (the timed region: every event
        costs a few microseconds of stream time, so only the dominant kernel is bracketed there); "all" = events
        around every kernel (the instrumented pass after the timed region)."""
        if args.graph and timed != "all":
            graphs[i].replay()
        elif timed == "all":
            ev = [new_event() for _ in range(4)]
            launch(i, ev)
            ev.append(new_event())
            ev_kernels.append(ev)
        elif timed:
            ev = [None, None, new_event(), new_event()]
            launch(i, ev)
            ev_grad.append((ev[2], ev[3]))
        else:
            launch(i)
        if multi:
            if on_device_comm:  # RCCL, enqueued behind the step; awaited before the segment's closing barrier
                pending.append(dist.all_reduce(scal[i], async_op=True))
            else:  # gloo rehearsal: reduce a host copy
                host = scal[i].cpu()
                dist.all_reduce(host)

    # barrier = a 1-element all-reduce (on the device for RCCL, on the host for gloo rehearsals)
    bar = torch.zeros(1, dtype=torch.float32, device=comm_dev)

    def fence():
        for w in pending:
            w.wait()
        pending.clear()
        torch.cuda.synchronize(dev)
        if multi:
            dist.all_reduce(bar)
            torch.cuda.synchronize(dev)

    def run_steps(n_steps, timed):
        """n_steps steps in segments of at most ring_n; returns the seconds between the fences."""
        elapsed, done = 0.0, 0
        while done < n_steps:
            refill()
            n = min(ring_n, n_steps - done)
            fence()
            t0 = time.perf_counter()
            for i in range(n):
                step(i, timed)
            fence()
            elapsed += time.perf_counter() - t0
            done += n
        return elapsed

    dbg(f"buffers ready (ring {ring_n})")
    if args.graph:
        refill()
        launch(0)  # load the kernels outside the capture
        torch.cuda.synchronize(dev)
        try:
            for i in range(ring_n):
                g = torch.cuda.CUDAGraph()
                # thread-local: other threads of this process (the collective library's watchdog) may go on making
                # HIP calls while this thread captures
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    launch(i, on=torch.cuda.current_stream(dev))
                graphs[i] = g
        except Exception as e:  # no graph: run the steps eagerly
            print(f"bench.py: HIP graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graphs.clear()
            args.graph = False
            torch.cuda.synchronize(dev)
    run_steps(args.warmup, False)
    dbg("warmup done")
    elapsed_local = run_steps(args.steps, True)
    dbg("timed region done")

    def reduce_max(v):
        t = torch.tensor([v], dtype=torch.float64, device=comm_dev)
        if multi:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather(vals):
        t = torch.tensor([vals], dtype=torch.float64, device=comm_dev)
        if world == 1:
            return t.cpu().numpy()
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return torch.cat(out).cpu().numpy()

    elapsed = reduce_max(elapsed_local)
    per_rank = gather([elapsed_local / args.steps * 1e3, cells, padded_rows, B, T, U1])
    total_cells = float(per_rank[:, 1].sum())

    flags = ctypes.c_int32(0)
    rc = L.rnnt_mi355x_read_status(_p(ws), ctypes.byref(flags), sp)
    if rc != 0:
        raise RuntimeError(f"device-side validation failed: flags {flags.value}")
    if not bool(torch.isfinite(costs).all()):
        raise RuntimeError("non-finite costs")
    n_cells_seen, n_pruned = _lib.read_stats(_p(ws), sp)
    if n_cells_seen != cells:
        raise RuntimeError(f"the gradient pass visited {n_cells_seen} lattice cells, expected {cells}")

    run_steps(min(ring_n, 6), "all")  # instrumented pass outside the timed region: every kernel bracketed

    ms_per_step = elapsed / args.steps * 1e3
    value = total_cells * args.steps / elapsed
    bpc = bytes_per_cell(V)
    kb = kernel_bytes_per_cell(V)
    kt = np.array([[e[k].elapsed_time(e[k + 1]) for k in range(4)] for e in ev_kernels])  # lse, sweep, grad, scalars
    step_dev = np.array([e[0].elapsed_time(e[4]) for e in ev_kernels])
    k_ms = kt.mean(axis=0)
    if ev_grad:  # the dominant kernel's time comes from the timed region itself
        k_ms[2] = float(np.mean([a.elapsed_time(b) for a, b in ev_grad]))
    pruned_frac = n_pruned / max(cells, 1)

    counter = counter_bytes_for(args.config, B, prune_log2) if world == 1 and args.logit_scale == 1.0 else {}

    def kernel_entry(name, ms, alg_bytes, traffic_model, key):
        e = {"name": name, "ms": float(ms), "algorithmic_bytes": int(alg_bytes),
             "achieved_GBs": alg_bytes / (ms * 1e-3) / 1e9, "frac_of_peak": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "traffic_model_bytes": int(traffic_model), "counter_bytes": counter.get(key)}
        return e

    row = 4 * V
    kernels = [
        kernel_entry("k_lse", k_ms[0], cells * kb["k_lse"], cells * (row + 16), "k_lse_hbm_bytes_per_launch"),
        kernel_entry("k_sweep", k_ms[1], cells * kb["k_sweep"], cells * 32, "k_sweep_hbm_bytes_per_launch"),
        kernel_entry("k_grad", k_ms[2], cells * kb["k_grad"],
                     padded_rows * row + (cells - n_pruned) * row + cells * 40, "k_grad_hbm_bytes_per_launch"),
    ]
    dominant = max(kernels, key=lambda k: k["ms"])
    d2d = d2d_copy_rate()
    step_frac = value / world * bpc / (HBM_PEAK_GBS * 1e9)

    if sharded:
        how = (f"the configuration's one global batch (B={cfg['B']}) sharded by utterance over {world} rank(s), "
               f"partition '{args.partition}', each rank padded to its own longest T and U+1")
    else:
        how = f"a full batch (B={B}) per rank: global batch = {B}*n_gpus"
    out = {
        "metric": "RNN-T loss fwd+bwd lattice cells/sec",
        "value": value,
        "unit": "cells/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if sharded else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "backend": (args.backend if multi else None),
        "config": {
            "workload": f"config {args.config}: {cfg['name']}{' (B overridden to %d)' % cfg['B'] if args.batch else ''}, "
                        f"fp32 logits ~ N(0,1){'*%g' % args.logit_scale if args.logit_scale != 1.0 else ''}, "
                        f"blank={blank}, grad_scale=1/B; {how}",
            "global_batch": cfg["B"] if sharded else B * world, "total_cells": int(total_cells),
            "ring_buffers": ring_n, "prune_log2": prune_log2, "hip_graph": bool(args.graph),
            "parallelism": (f"utterance-sharded dp{world}, one 16-byte all-reduce per step" if world > 1 else "single GPU"),
        },
        "ranks_ms_per_step": {"min": float(per_rank[:, 0].min()), "median": float(np.median(per_rank[:, 0])),
                              "max": float(per_rank[:, 0].max())},
        "per_rank": [{"rank": r, "ms_per_step": float(v[0]), "cells": int(v[1]), "padded_rows": int(v[2]),
                      "utterances": int(v[3]), "maxT": int(v[4]), "maxU1": int(v[5])} for r, v in enumerate(per_rank)],
        "pruned_row_frac": pruned_frac,
        "step_ms_device": {"median": float(np.median(step_dev)), "min": float(step_dev.min()), "max": float(step_dev.max())},
        "d2d_copy_GBs": d2d,
        "step_frac_of_d2d_copy": value / world * bpc / (d2d * 1e9),
        # whole-step accounting: every cell's algorithmic bytes (8V+16) over the wall-clock step, per GPU
        "roofline": {
            "bound": "hbm", "scope": "whole step (k_lse + k_sweep + k_grad + scalars), rank 0's GPU",
            "achieved": value / world * bpc / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_frac,
            "algorithmic_bytes_per_cell": bpc,
            "traffic": counter.get("step_hbm_bytes"),
            "traffic_source": ("profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of this "
                               "workload; not measured in this run)" if counter else None),
            "dominant_kernel": dominant["name"],
            "kernel_times_from": ("k_grad: HIP events on the launch stream around every launch of the timed region; k_lse, "
                                  "k_sweep and step_ms_device: an instrumented pass of %d steps right after it"
                                  % min(ring_n, 6) if not args.graph else
                                  "HIP events in an instrumented pass right after the timed region (which replays graphs)"),
            "kernels": kernels,
        },
    }
    solo = rank == 0 and world == 1
    if solo and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(master, labels, il_h, ll_h, cfg)
        out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    if solo and not args.no_legs:
        legs = {}

        def leg(name, what):
            sec = run_steps(args.leg_steps, False)
            c, p = _lib.read_stats(_p(ws), sp)
            v = cells * args.leg_steps / sec
            legs[name] = {"what": what, "ms_per_step": sec / args.leg_steps * 1e3, "value": v,
                          "frac": v * bpc / (HBM_PEAK_GBS * 1e9), "pruned_row_frac": p / max(c, 1), "steps": args.leg_steps}

        if args.graph:
            args.graph = False
        call_prune[0] = float("-inf")
        leg("prune_off", "same inputs, pruning disabled: every row is read twice and written once")
        call_prune[0] = _lib.prune_arg(prune_log2)
        master.mul_(10.0 / args.logit_scale)
        leg("peaked", "logits ~ N(0,1)*10 (posteriors as peaked as a trained model's), same pruning threshold")
        out["legs"] = legs
    if rank == 0:
        print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        from rnnt_amd import build as _build   # once, here: the ranks would otherwise all find it stale together
        if not os.environ.get("RNNT_MI355X_LIB") and os.path.exists(_build.LIB) and _build.needs_build():
            try:
                _build.build()
            except Exception as e:
                print(f"bench.py: librnnt_mi355x.so is older than its sources and could not be rebuilt ({e})", file=sys.stderr)
        raise SystemExit(spawn_ranks(args.gpus, argv))
    run_rank(args)


if __name__ == "__main__":
    main()
