#!/usr/bin/env python3
"""Fused additive-joint loss + gradients (rnnt_joint_loss, forward + backward) at B=32, T=400, U=100 over a range
of vocabularies: ms per step and the three contractions' rate.  Shows whether any vocabulary falls off the curve."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rnnt_amd
dev = torch.device("cuda:0")
B, T, U = 32, 400, 100
for V in (int(a) for a in (sys.argv[1:] or "32 40 62 64 100 128 129 200 256 300 512 1000 1024 2048 4096".split())):
    f = torch.randn(B, T, V, device=dev, requires_grad=True)
    g = torch.randn(B, U + 1, V, device=dev, requires_grad=True)
    lab = torch.randint(0, V - 1, (B, U), device=dev, dtype=torch.int32)
    il = torch.full((B,), T, dtype=torch.int32); ll = torch.full((B,), U, dtype=torch.int32)
    def step():
        f.grad = g.grad = None
        rnnt_amd.rnnt_joint_loss(f, g, lab, il, ll, V - 1, check_inputs=False).mean().backward()
    for _ in range(3): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): step()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    flop = 3 * 2.0 * B * T * (U + 1) * V
    print(f"V={V:5d}  {ms:7.3f} ms  {B * T * (U + 1) / ms / 1e3:8.1f} Mcells/s  contractions {flop / ms / 1e9:6.1f} TFLOP/s", flush=True)
