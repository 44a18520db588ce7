#!/usr/bin/env python3
"""Wall time per training-style call of the Python layer (autograd forward + backward) against the device time of the
same work, at the reference's own TIMIT shape (BASELINE config 1) and at config 2."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rnnt_amd
dev = torch.device("cuda:0")
for name, (B, T, U, V) in {"config 1": (8, 150, 40, 62), "config 2": (32, 500, 100, 62)}.items():
    base = torch.randn(B, T, U + 1, V, device=dev)
    lab = torch.randint(0, 39, (B, U), dtype=torch.int32)           # host tensors, as a data loader delivers them
    il = torch.full((B,), T, dtype=torch.int32); ll = torch.full((B,), U, dtype=torch.int32)
    lab_d, il_d, ll_d = lab.to(dev), il.to(dev), ll.to(dev)
    def step(labels, i, l, **kw):
        acts = base.clone().requires_grad_(True)
        rnnt_amd.rnnt_loss(acts, labels, i, l, 39, **kw).mean().backward()
    for label, args, kw in (("host lengths/labels", (lab, il, ll), {}), ("device tensors, check_inputs", (lab_d, il_d, ll_d), {}),
                            ("device tensors, no check", (lab_d, il_d, ll_d), {"check_inputs": False})):
        for _ in range(5): step(*args, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 200
        for _ in range(n): step(*args, **kw)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"{name}: {label:32s} {dt*1e6:8.1f} us per step (clone + loss + mean + backward)")
