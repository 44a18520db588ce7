#!/usr/bin/env python3
"""Runs the forward (lse + sweeps) at T=500 with U+1 = 64, 101, 200, 400 (1, 2, 4, 8 lattice columns per lane) on a
tiny vocabulary, for a rocprofv3 --kernel-trace --stats pass: time per diagonal of k_sweep<K>."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnt_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0"); p = lambda t: ctypes.c_void_p(t.data_ptr())
B, T, V = (int(sys.argv[1]) if len(sys.argv) > 1 else 32), 500, 8   # usage: sweep_probe.py [B]
for U in (63, 100, 199, 399):
    x = torch.randn(B, T, U + 1, V, device=dev)
    lab = torch.randint(1, V, (B, U), device=dev, dtype=torch.int32)
    il = torch.full((B,), T, dtype=torch.int32, device=dev); ll = torch.full((B,), U, dtype=torch.int32, device=dev)
    costs = torch.empty(B, device=dev)
    ws = torch.empty(_lib.workspace_bytes(B, T, U + 1), dtype=torch.uint8, device=dev)
    for _ in range(5):
        assert L.rnnt_mi355x_forward(p(x), p(lab), p(il), p(ll), B, T, U + 1, V, 0, p(costs), p(ws), None) == 0
    torch.cuda.synchronize()
    print(U, costs[:2].tolist())
