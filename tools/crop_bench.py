#!/usr/bin/env python3
"""Times rnnt_lambda_func_v2 (crop of 2 frames inside the library) against the same loss on a copied crop, forward +
backward, at the reference's FC-joint shape (B=32, T0=502, U=100, V=62) and at V=1024 (B=16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rnnt_amd
dev = torch.device("cuda:0")
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for B, T0, U, V in ((32, 502, 100, 62), (16, 402, 100, 1024)):
    acts = torch.randn(B, T0, U + 1, V, device=dev, requires_grad=True)
    lab = torch.randint(0, 39, (B, U), device=dev, dtype=torch.int32)
    il = torch.full((B, 1), T0, dtype=torch.int32, device=dev); ll = torch.full((B, 1), U, dtype=torch.int32, device=dev)
    def in_library():
        acts.grad = None
        rnnt_amd.rnnt_lambda_func_v2([acts, lab, il, ll]).mean().backward()
    def copied():
        acts.grad = None
        crop = acts[:, 2:].contiguous()
        rnnt_amd.rnnt_loss(crop, lab, (il - 2).reshape(B), ll.reshape(B), 39, check_inputs=True).mean().backward()
    print(f"B={B} T0={T0} U={U} V={V}: crop inside the library {timeit(in_library):.3f} ms | copied crop {timeit(copied):.3f} ms")
