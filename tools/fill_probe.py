#!/usr/bin/env python3
"""k_fill_rows alone on buffers allocated the way bench.py's config 5 allocates them (experiment build with
-DRNNT_RMW_TIMING).  usage: RNNT_MI355X_LIB=tools/variants/rmwt.so fill_probe.py [n_buffers]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnnt_amd import _lib

L = _lib.lib()
f = L.rnnt_mi355x_debug_fill
f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
nbuf = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B, T, U1, V = 128, 1000, 200, 512
dev = torch.device("cuda:0")
rows = T * U1
bits = torch.zeros(B * ((rows + 63) // 64 * 2), dtype=torch.int32, device=dev)
bufs = [torch.empty((B, T, U1, V), device=dev) for _ in range(nbuf)]
gb = bufs[0].numel() * 4 / 1e9
def timed(fn, n=3):
    best = 1e9
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
st = torch.cuda.current_stream().cuda_stream
for i, x in enumerate(bufs):
    ms = timed(lambda: f(x.data_ptr(), x[0].numel(), rows, V // 256, bits.data_ptr(), B, st))
    print(f"buffer {i} ({gb:.1f} GB) untouched: k_fill_rows {ms:.3f} ms = {gb / ms * 1e3:.0f} GB/s")
for i, x in enumerate(bufs):
    x.normal_()
    torch.cuda.synchronize()
    ms = timed(lambda: f(x.data_ptr(), x[0].numel(), rows, V // 256, bits.data_ptr(), B, st), n=1)
    ms2 = timed(lambda: f(x.data_ptr(), x[0].numel(), rows, V // 256, bits.data_ptr(), B, st), n=1)
    print(f"buffer {i} after normal_(): first fill {ms:.3f} ms = {gb / ms * 1e3:.0f} GB/s, second {ms2:.3f} ms")
    ms = timed(lambda: x.zero_(), n=1)
    print(f"buffer {i} torch zero_(): {ms:.3f} ms = {gb / ms * 1e3:.0f} GB/s")
# every second frame masked out (runs of 200 rows filled, 200 skipped)
m = torch.zeros(B, (rows + 63) // 64 * 64, dtype=torch.bool, device=dev)
m[:, :rows].view(B, T, U1)[:, 1::2, :] = True
w = m.view(B, -1, 32).to(torch.int64)
packed = (w << torch.arange(32, device=dev, dtype=torch.int64)).sum(-1)
bits2 = packed.to(torch.int32).contiguous() if packed.max() < 2**31 else (packed - (packed >= 2**31) * 2**32).to(torch.int32).contiguous()
x = bufs[0]
ms = timed(lambda: f(x.data_ptr(), x[0].numel(), rows, V // 256, bits2.data_ptr(), B, st))
print(f"every second frame masked: {ms:.3f} ms = {gb / 2 / ms * 1e3:.0f} GB/s of zeros")
