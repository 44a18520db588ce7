#!/usr/bin/env python3
"""Why is a zero-fill slower over a buffer that holds random data?  (experiment build with -DRNNT_RMW_TIMING)"""
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
def fill(x):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); f(x.data_ptr(), x[0].numel(), rows, V // 256, bits.data_ptr(), B, st); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
def show(tag, ms): print(f"{tag:60s} {ms:7.3f} ms = {gb / ms * 1e3:5.0f} GB/s", flush=True)
show("zeros over zeros (a)", fill(a)); show("zeros over zeros (a) again", fill(a))
a.normal_(); torch.cuda.synchronize()
show("a.normal_(), then fill b (zeros)", fill(b))
show("then fill a (random data)", fill(a))
show("then fill a again", fill(a))
a.fill_(1.0); torch.cuda.synchronize()
show("a.fill_(1.0), then fill a", fill(a))
a.normal_(); torch.cuda.synchronize(); time.sleep(0.5)
show("a.normal_(), half a second's sleep, then fill a", fill(a))
a.normal_(); torch.cuda.synchronize()
c = a.clone(); torch.cuda.synchronize()
show("a.normal_(), c = a.clone(), then fill c", fill(c))
a.uniform_(0.0, 1e-30); torch.cuda.synchronize()
show("a.uniform_(0, 1e-30) (same exponent bits mostly), then fill a", fill(a))
ai = a.view(torch.int32); ai.random_(0, 2); torch.cuda.synchronize()
show("words 0 or 1 at random, then fill a", fill(a))
ai.random_(0, 256); torch.cuda.synchronize()
show("random low byte only, then fill a", fill(a))
ai.random_(-2**31, 2**31 - 1); torch.cuda.synchronize()
show("all 32 bits random, then fill a", fill(a))
x1 = torch.full_like(a, 1.0)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
a.zero_(); torch.cuda.synchronize(); e0.record(); a.copy_(x1); e1.record(); torch.cuda.synchronize()
print(f"copy of constants over zeros: {e0.elapsed_time(e1):.3f} ms")
x1.normal_(); a.zero_(); torch.cuda.synchronize(); e0.record(); a.copy_(x1); e1.record(); torch.cuda.synchronize()
print(f"copy of random data over zeros: {e0.elapsed_time(e1):.3f} ms")
