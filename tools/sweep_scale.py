#!/usr/bin/env python3
"""Time per anti-diagonal of the lattice sweep against batch size, frame count and label count (tiny vocabulary, so
that only the sweep matters): events around k_sweep from rnnt_mi355x_loss_grad_events.
usage: sweep_scale.py [U ...]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnt_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0"); p = lambda t: ctypes.c_void_p(t.data_ptr())
V = 8
Us = [int(a) for a in sys.argv[1:]] or [100, 199]
for U in Us:
    for T in (500, 1000):
        for B in (8, 16, 32, 64, 128, 256):
            x = torch.randn(B, T, U + 1, V, device=dev)
            lab = torch.randint(1, V, (B, U), device=dev, dtype=torch.int32)
            il = torch.full((B,), T, dtype=torch.int32, device=dev); ll = torch.full((B,), U, dtype=torch.int32, device=dev)
            costs = torch.empty(B, device=dev); gs = torch.ones(B, device=dev)
            ws = torch.empty(_lib.workspace_bytes(B, T, U + 1), dtype=torch.uint8, device=dev)
            st = torch.cuda.current_stream(dev)
            ts = []
            for _ in range(6):
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                for e in ev: e.record(st)
                arr = (ctypes.c_void_p * 4)(*[ctypes.c_void_p(e.cuda_event) for e in ev])
                rc = L.rnnt_mi355x_loss_grad_events(p(x), p(lab), p(il), p(ll), B, T, U + 1, V, 0, p(gs), p(costs), p(ws),
                                                    ctypes.c_void_p(st.cuda_stream), float("nan"), arr)
                assert rc == 0
                torch.cuda.synchronize()
                ts.append(ev[1].elapsed_time(ev[2]))
            ms = float(np.median(ts[1:]))
            print(f"U={U} T={T} B={B}: sweep {ms*1e3:7.1f} us  {ms*1e6/(T+U):6.1f} ns/diagonal", flush=True)
            del x, ws
