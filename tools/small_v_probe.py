#!/usr/bin/env python3
"""Forward + in-place backward for a list of shapes, for a rocprofv3 --kernel-trace pass (tools/small_v_stats.sh):
bandwidth of the two streaming passes.  Arguments: vocabularies V (at B=64, T=500, U=100) or B:T:U:V tuples."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnt_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0"); p = lambda t: ctypes.c_void_p(t.data_ptr())
for a in (sys.argv[1:] or ["32", "40", "62", "64"]):
    B, T, U, V = (int(v) for v in a.split(":")) if ":" in a else (64, 500, 100, int(a))
    master = torch.randn(B, T, U + 1, V, device=dev)
    x = torch.empty_like(master)
    lab = torch.randint(1, V, (B, U), device=dev, dtype=torch.int32)
    il = torch.full((B,), T, dtype=torch.int32, device=dev); ll = torch.full((B,), U, dtype=torch.int32, device=dev)
    costs = torch.empty(B, device=dev)
    ws = torch.empty(_lib.workspace_bytes(B, T, U + 1), dtype=torch.uint8, device=dev)
    for _ in range(5):
        x.copy_(master)   # the gradient overwrites the logits: same input every time
        assert L.rnnt_mi355x_loss_grad(p(x), p(lab), p(il), p(ll), B, T, U + 1, V, 0, None, p(costs), p(ws), None, float("nan")) == 0
    torch.cuda.synchronize()
    print(B, T, U, V, x.numel() * 4 / 1e6, "MB")
