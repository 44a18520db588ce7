#!/usr/bin/env python3
"""How much cached traffic before a zero-fill makes it slow, and what resets it.  (experiment build, -DRNNT_RMW_TIMING)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnnt_amd import _lib

L = _lib.lib()
f = L.rnnt_mi355x_debug_fill
f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
B, T, U1, V = 32, 1000, 200, 512
dev = torch.device("cuda:0")
rows = T * U1
bits = torch.zeros(B * ((rows + 63) // 64 * 2), dtype=torch.int32, device=dev)
a = torch.zeros((B, T, U1, V), device=dev); b = torch.zeros((B, T, U1, V), device=dev)
gb = a.numel() * 4 / 1e9
st = torch.cuda.current_stream().cuda_stream
def fill(x, nb=B):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); f(x.data_ptr(), x[0].numel(), rows, V // 256, bits.data_ptr(), nb, st); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
def show(tag, ms, g=gb): print(f"{tag:72s} {ms:7.3f} ms = {g / ms * 1e3:5.0f} GB/s", flush=True)
fill(a); fill(a)
show("fill a (warm)", fill(a))
flat = b.view(-1)
for mb in (1, 4, 16, 64, 256, 1024):
    n = mb * (1 << 20) // 4
    s = flat[:n].sum(); torch.cuda.synchronize()
    show(f"sum over {mb} MB of b, then fill a", fill(a))
    show("   fill a again", fill(a))
s = flat[: (1 << 30) // 4].sum(); torch.cuda.synchronize()
show("sum over 1 GB of b, then fill the first utterance of b (0.41 GB)", fill(b, 1), gb / B)
show("   then fill a", fill(a))
s = flat[: (1 << 30) // 4].sum(); torch.cuda.synchronize()
small = torch.zeros(1024, device=dev); small.add_(1.0); torch.cuda.synchronize()
show("sum over 1 GB of b, a tiny torch kernel, then fill a", fill(a))
s = flat[: (1 << 30) // 4].sum(); torch.cuda.synchronize(); time.sleep(0.3)
show("sum over 1 GB of b, 0.3 s idle, then fill a", fill(a))
# the slow fill: where is the time?  four quarters as separate launches
s = flat[: (1 << 30) // 4].sum(); torch.cuda.synchronize()
for q in range(4):
    show(f"sum over 1 GB of b, then quarter {q} of a", fill(a[q * 8:], 8), gb / 4)
