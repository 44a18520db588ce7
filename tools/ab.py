#!/usr/bin/env python3
"""Interleaved A/B timing of several builds of librnnt_mi355x.so in ONE process.
usage: tools/ab.py [--config N] [--rounds R] lib1.so lib2.so ...
Reports median/min of forward (k_init+k_lse+k_sweep) and backward (k_grad) per build."""
import argparse, ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CONFIGS, make_lengths  # noqa
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
cfg = dict(CONFIGS[a.config])
if a.batch: cfg["B"] = a.batch
B, T, U, V, blank = cfg["B"], cfg["T"], cfg["U"], cfg["V"], cfg["blank"]
U1 = U + 1
dev = torch.device("cuda:0")
il_h, ll_h = make_lengths(cfg, 0, 0xC0FFEE + a.config)
master = torch.empty((B, T, U1, V), dtype=torch.float32, device=dev).normal_()
buf = torch.empty_like(master)
labels = torch.randint(0, V - 1, (B, U), device=dev, dtype=torch.int32)
il = torch.from_numpy(il_h).to(dev); ll = torch.from_numpy(ll_h).to(dev)
gs = torch.full((B,), 1.0 / B, device=dev)
costs = torch.empty(B, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
libs = []
for path in a.libs:
    L = ctypes.CDLL(os.path.abspath(path))
    n = ctypes.c_size_t(0)
    L.rnnt_mi355x_workspace_bytes(B, T, U1, ctypes.byref(n))
    ws = torch.empty(n.value, dtype=torch.uint8, device=dev)
    libs.append((os.path.basename(path), L, ws))
meta = (B, T, U1, V, blank)
res = {n: ([], []) for n, _, _ in libs}
ref = None
fused = {}
for r in range(a.rounds + 1):
    for name, L, ws in libs:
        buf.copy_(master)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        rc = L.rnnt_mi355x_forward(p(buf), p(labels), p(il), p(ll), *meta, p(costs), p(ws), sp)
        e[1].record()
        rc |= L.rnnt_mi355x_backward_inplace(p(buf), p(labels), p(il), p(ll), *meta, p(gs), p(ws), sp, float("nan"))
        e[2].record()
        torch.cuda.synchronize()
        assert rc == 0
        if hasattr(L, "rnnt_mi355x_loss_grad_events") and r > 0:
            buf.copy_(master)
            f0 = torch.cuda.Event(enable_timing=True); f1 = torch.cuda.Event(enable_timing=True)
            f0.record()
            rc = L.rnnt_mi355x_loss_grad(p(buf), p(labels), p(il), p(ll), *meta, p(gs), p(costs), p(ws), sp, float("nan"))
            f1.record(); torch.cuda.synchronize(); assert rc == 0
            fused.setdefault(name, []).append(f0.elapsed_time(f1))
        if r == 0:
            c = costs.clone(); g = buf[0, :2].clone()
            if ref is None: ref = (c, g)
            else: print(name, "max |dcost|", (c - ref[0]).abs().max().item(), "max |dgrad|", (g - ref[1]).abs().max().item())
            continue
        res[name][0].append(e[0].elapsed_time(e[1])); res[name][1].append(e[1].elapsed_time(e[2]))
cells = int((il_h.astype(np.int64) * (ll_h + 1)).sum())
for name, (f, bk) in res.items():
    tot = np.median(f) + np.median(bk)
    if name in fused: print(f"{name:22s} fused loss_grad med {np.median(fused[name]):7.3f} min {min(fused[name]):7.3f} ms  frac {cells*(8*V+16)/np.median(fused[name])/1e-3/8e12:.3f}")
    print(f"{name:22s} fwd med {np.median(f):7.3f} min {min(f):7.3f} | bwd med {np.median(bk):7.3f} min {min(bk):7.3f} | sum {tot:7.3f} ms  {cells/tot/1e6:8.1f} Mcells/s  frac {cells*(8*V+16)/tot/1e-3/8e12:.3f}")
