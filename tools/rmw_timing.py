#!/usr/bin/env python3
"""Per-phase cycle sums of k_grad_rmw from an experiment build (-DRNNT_RMW_TIMING): one loss+gradient at the
config-3 shape, N(0,1) logits.  usage: RNNT_MI355X_LIB=tools/variants/<name>.so rmw_timing.py [scale] [prune_log2]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnnt_amd import _lib

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
prune = float(sys.argv[2]) if len(sys.argv) > 2 else float("nan")
L = _lib.lib()
f = L.rnnt_mi355x_debug_rmw_timing
f.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
B, T, U1, V = 64, 400, 101, 1024
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.empty((B, T, U1, V), device=dev).normal_(generator=g).mul_(scale)
lab = torch.randint(0, V - 1, (B, U1 - 1), generator=g, device=dev, dtype=torch.int32)
il = torch.full((B,), T, dtype=torch.int32, device=dev); ll = torch.full((B,), U1 - 1, dtype=torch.int32, device=dev)
costs = torch.empty(B, device=dev); ws = torch.empty(_lib.workspace_bytes(B, T, U1), dtype=torch.uint8, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
buf = (ctypes.c_ulonglong * 8)()
spans = (ctypes.c_ulonglong * 2048)()
fs = L.rnnt_mi355x_debug_rmw_spans
fs.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
for rep in range(3):
    xx = x.clone()
    torch.cuda.synchronize()
    f(buf)
    assert L.rnnt_mi355x_loss_grad(p(xx), p(lab), p(il), p(ll), B, T, U1, V, V - 1, None, p(costs), p(ws), None, prune) == 0
    f(buf)
    v = list(buf)
    n = max(v[7], 1)
    names = ["first entries", "issue loads", "barrier", "loads arrive", "write phase", "closing"]
    print(f"items {v[7]}  per item (cycles of the 100 MHz counter x10 = ns): " +
          "  ".join(f"{nm} {v[i] / n * 10:.0f}" for i, nm in enumerate(names)))
    fs(spans)
    st = [spans[i] for i in range(512)]; en = [spans[1024 + i] for i in range(512)]
    t0 = min(st)
    en_s = sorted((e - t0) / 100.0 for e in en)
    print(f"  blocks start {min(st)-t0:.0f}..{(max(st)-t0)/100:.1f} us; end: first {en_s[0]:.1f} us, 10% {en_s[51]:.1f}, median {en_s[256]:.1f}, 90% {en_s[460]:.1f}, last {en_s[-1]:.1f}")
    early = sum(1 for x in st if (x - t0) / 100.0 < 20.0)
    dur = sorted((e - b) / 100.0 for b, e in zip(st, en))
    print(f"  blocks started within 20 us: {early} of 512; block duration min {dur[0]:.0f} / median {dur[256]:.0f} / max {dur[-1]:.0f} us")
    late = sorted((x - t0) / 100.0 for x in st if (x - t0) / 100.0 >= 20.0)
    if late: print(f"  late starts: first {late[0]:.0f} us, median {late[len(late)//2]:.0f}, last {late[-1]:.0f}")
