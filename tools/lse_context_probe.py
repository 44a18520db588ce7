#!/usr/bin/env python3
"""Does the first pass run at a different pace depending on what ran before it?  Under rocprofv3 --kernel-trace:
full steps (the first pass follows the previous step's zero-fill), then forward-only calls (it follows a sweep),
then pairs of forward calls on alternating buffers."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnnt_amd import _lib

L = _lib.lib()
B, T, U1, V = 64, 400, 101, 1024
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
master = torch.empty((B, T, U1, V), device=dev).normal_(generator=g)
ring = [master.clone() for _ in range(6)]
lab = torch.randint(0, V - 1, (B, U1 - 1), generator=g, device=dev, dtype=torch.int32)
il = torch.full((B,), T, dtype=torch.int32, device=dev); ll = torch.full((B,), U1 - 1, dtype=torch.int32, device=dev)
gs = torch.full((B,), 1.0 / B, device=dev)
costs = torch.empty(B, device=dev); ws = torch.empty(_lib.workspace_bytes(B, T, U1), dtype=torch.uint8, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
def full(x):
    assert L.rnnt_mi355x_loss_grad(p(x), p(lab), p(il), p(ll), B, T, U1, V, V - 1, p(gs), p(costs), p(ws), None, float("nan")) == 0
def fwd(x):
    assert L.rnnt_mi355x_forward(p(x), p(lab), p(il), p(ll), B, T, U1, V, V - 1, p(costs), p(ws), None) == 0
for x in ring[:4]: full(x)
torch.cuda.synchronize()
for x in ring[:4]: x.copy_(master)
torch.cuda.synchronize()
for x in ring[4:] + ring[:4]: fwd(x)       # 6 forward-only calls
torch.cuda.synchronize()
for x in ring[:4]: full(x)                  # 4 full steps again
torch.cuda.synchronize()
