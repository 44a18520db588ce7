#!/usr/bin/env python3
"""One lattice shape through the forward (first pass + sweeps), a few times: for a kernel profile.  usage: B T U"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnt_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0"); p = lambda t: ctypes.c_void_p(t.data_ptr())
B, T, U = (int(a) for a in sys.argv[1:4]); V = 8
x = torch.randn(B, T, U + 1, V, device=dev)
lab = torch.randint(1, V, (B, U), device=dev, dtype=torch.int32)
il = torch.full((B,), T, dtype=torch.int32, device=dev); ll = torch.full((B,), U, dtype=torch.int32, device=dev)
costs = torch.empty(B, device=dev)
ws = torch.empty(_lib.workspace_bytes(B, T, U + 1), dtype=torch.uint8, device=dev)
for _ in range(6):
    assert L.rnnt_mi355x_forward(p(x), p(lab), p(il), p(ll), B, T, U + 1, V, 0, p(costs), p(ws), None) == 0
torch.cuda.synchronize()
