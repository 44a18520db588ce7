#!/usr/bin/env python3
"""Runs the forward (first pass + sweeps) on dense batches of several vocabulary sizes and on a ragged one, for a
rocprofv3 kernel trace: tools/lse_v_trace.sh prints the first pass's time and rate per case."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnt_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0"); p = lambda t: ctypes.c_void_p(t.data_ptr())
B, T, U = 32, 250, 100
g = torch.Generator(device=dev); g.manual_seed(5)
for V, ragged in ((132, 0), (200, 0), (300, 0), (512, 0), (512, 1), (203, 0), (1030, 0)):
    x = torch.randn(B, T, U + 1, V, device=dev)
    lab = torch.randint(1, V, (B, U), device=dev, dtype=torch.int32)
    if ragged:
        il = torch.randint(T // 4, T + 1, (B,), generator=g, device=dev, dtype=torch.int32)
        ll = torch.randint(U // 10, U + 1, (B,), generator=g, device=dev, dtype=torch.int32)
        il[0] = T; ll[0] = U
    else:
        il = torch.full((B,), T, dtype=torch.int32, device=dev); ll = torch.full((B,), U, dtype=torch.int32, device=dev)
    costs = torch.empty(B, device=dev)
    ws = torch.empty(_lib.workspace_bytes(B, T, U + 1), dtype=torch.uint8, device=dev)
    for _ in range(5):
        assert L.rnnt_mi355x_forward(p(x), p(lab), p(il), p(ll), B, T, U + 1, V, 0, p(costs), p(ws), None) == 0
    torch.cuda.synchronize()
    cells = int((il.long() * (ll.long() + 1)).sum())
    print("CASE", V, ragged, cells, flush=True)
