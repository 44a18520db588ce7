#!/usr/bin/env python3
"""What makes the first zero-fill after another kernel slow: the other kernel's pages (address translation), an
idle gap (clocks), or the switch itself?  (experiment build with -DRNNT_RMW_TIMING)"""
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
small = torch.zeros(1024, device=dev)
gb = a.numel() * 4 / 1e9
st = torch.cuda.current_stream().cuda_stream
def fill(x):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); f(x.data_ptr(), x[0].numel(), rows, V // 256, bits.data_ptr(), B, st); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
def show(tag, ms): print(f"{tag:64s} {ms:7.3f} ms = {gb / ms * 1e3:5.0f} GB/s", flush=True)
fill(a); fill(a)
show("fill a (warm)", fill(a))
small.add_(1.0); torch.cuda.synchronize()
show("a tiny torch kernel, then fill a", fill(a))
time.sleep(0.5)
show("half a second idle, then fill a", fill(a))
show("fill a", fill(a))
show("fill b (other buffer, same kernel)", fill(b))
show("fill a", fill(a))
show("fill a", fill(a))
s = a.sum(); torch.cuda.synchronize()
show("a.sum() (reads all of a), then fill a", fill(a))
s = b.sum(); torch.cuda.synchronize()
show("b.sum() (reads all of b), then fill a", fill(a))
show("fill a", fill(a))
s = b[:4].sum(); torch.cuda.synchronize()
show("b[:4].sum() (reads 1.6 GB of b), then fill a", fill(a))
a.fill_(1.0); torch.cuda.synchronize()
show("a.fill_(1.0), then fill a", fill(a))
a.fill_(0.0); torch.cuda.synchronize()
show("a.fill_(0.0), then fill a", fill(a))
