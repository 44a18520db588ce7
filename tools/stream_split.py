#!/usr/bin/env python3
"""Utterances are independent, so a caller can split a batch over several HIP streams (one workspace each): the
latency-bound lattice sweep of one part then overlaps the streaming passes of the others.  Times the fused
loss + gradient call at B=32, T=500, U=100, V=62 and at B=64, T=400, U=100, V=1024 with 1, 2, 4 and 8 parts."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnt_amd import _lib

dev = torch.device("cuda:0")
L = _lib.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr())
for B, T, U, V in ((32, 500, 100, 62), (64, 400, 100, 1024), (8, 400, 100, 1024)):
    U1 = U + 1
    master = torch.randn(B, T, U1, V, device=dev)
    x = torch.empty_like(master)
    lab = torch.randint(0, V - 1, (B, U), device=dev, dtype=torch.int32)
    il = torch.full((B,), T, dtype=torch.int32, device=dev); ll = torch.full((B,), U, dtype=torch.int32, device=dev)
    gs = torch.full((B,), 1.0 / B, device=dev); costs = torch.empty(B, device=dev)
    for parts in (1, 2, 4, 8):
        nb = B // parts
        streams = [torch.cuda.Stream(dev) for _ in range(parts)]
        wss = [torch.empty(_lib.workspace_bytes(nb, T, U1), dtype=torch.uint8, device=dev) for _ in range(parts)]
        def step():
            main = torch.cuda.current_stream(dev)
            x.copy_(master)   # the gradient overwrites the logits; the copy's time is subtracted below
            fork = torch.cuda.Event(); fork.record(main)
            for i, s in enumerate(streams):
                s.wait_event(fork)
                b0 = i * nb
                rc = L.rnnt_mi355x_loss_grad(p(x[b0]), p(lab[b0]), p(il[b0:]), p(ll[b0:]), nb, T, U1, V, V - 1, p(gs[b0:]),
                                             p(costs[b0:]), p(wss[i]), ctypes.c_void_p(s.cuda_stream), float("nan"))
                assert rc == 0
                e = torch.cuda.Event(); e.record(s); main.wait_event(e)
        for _ in range(3): step()   # the gradient overwrites x: values change, timing does not
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps): step()
        e1.record(); torch.cuda.synchronize()
        total = e0.elapsed_time(e1) / reps
        e0.record()
        for _ in range(reps): x.copy_(master)
        e1.record(); torch.cuda.synchronize()
        print(f"B={B} T={T} U={U} V={V}: {parts} part(s) {total - e0.elapsed_time(e1) / reps:.3f} ms", flush=True)
