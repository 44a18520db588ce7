#!/usr/bin/env python3
"""Times the fused additive-joint loss (rnnt_mi355x_joint_forward/backward) against the
materialised path (broadcast add -> rnnt_mi355x_loss_grad -> reductions over u and t) on one GPU."""
import argparse, ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CONFIGS
from rnnt_amd import _lib
ap = argparse.ArgumentParser(); ap.add_argument("--config", type=int, default=3); ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
cfg = CONFIGS[a.config]; B, T, U, V, blank = cfg["B"], cfg["T"], cfg["U"], cfg["V"], cfg["blank"]; U1 = U + 1
dev = torch.device("cuda:0"); L = _lib.lib(); p = lambda t: ctypes.c_void_p(t.data_ptr())
f = torch.randn(B, T, V, device=dev); g = torch.randn(B, U1, V, device=dev)
lab = torch.randint(0, V - 1, (B, U), device=dev, dtype=torch.int32)
il = torch.full((B,), T, dtype=torch.int32, device=dev); ll = torch.full((B,), U, dtype=torch.int32, device=dev)
gs = torch.full((B,), 1.0 / B, device=dev); costs = torch.empty(B, device=dev)
df = torch.empty_like(f); dg = torch.empty_like(g)
ws = torch.empty(_lib.joint_workspace_bytes(B, T, U1, V), dtype=torch.uint8, device=dev)
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
meta = (B, T, U1, V, blank)
def fused():
    rc = L.rnnt_mi355x_joint_forward(p(f), T * V, p(g), p(lab), p(il), p(ll), *meta, p(costs), p(ws), sp)
    rc |= L.rnnt_mi355x_joint_backward(p(f), T * V, p(g), p(lab), p(il), p(ll), *meta, p(gs), p(df), T * V, p(dg), p(ws), sp, float("nan"))
    assert rc == 0
ws2 = torch.empty(_lib.workspace_bytes(B, T, U1), dtype=torch.uint8, device=dev)
def materialised():
    acts = f.unsqueeze(2) + g.unsqueeze(1)
    rc = L.rnnt_mi355x_loss_grad(p(acts), p(lab), p(il), p(ll), *meta, p(gs), p(costs), p(ws2), sp, float("nan"))
    assert rc == 0
    return acts.sum(2), acts.sum(1)
def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps
fused(); c1 = costs.clone(); d1, d2 = df.clone(), dg.clone()
m1, m2 = materialised(); torch.cuda.synchronize()
print("max |dcost|", (c1 - costs).abs().max().item(), "max |d df|", (d1 - m1).abs().max().item(), "max |d dg|", (d2 - m2).abs().max().item())
tf, tm = timeit(fused), timeit(materialised)
cells = B * T * U1
print(f"config {a.config}: fused joint {tf:.3f} ms ({cells/tf/1e3:.0f} Mcells/s) | materialised add+loss_grad+reduce {tm:.3f} ms ({cells/tm/1e3:.0f} Mcells/s) | speed-up {tm/tf:.1f}x")
