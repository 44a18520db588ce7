#!/usr/bin/env python3
"""Forward time of the Dense layer (K, N pairs) through the C ABI, for each library variant given."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
M = 32 * 500 * 101
p = lambda t: ctypes.c_void_p(t.data_ptr())
for name in sys.argv[1:]:
    L = ctypes.CDLL(os.path.join(ROOT, "tools", "variants", name + ".so"))
    out = []
    for K, N in ((100, 62), (100, 100), (128, 128), (64, 40)):
        x = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev) * 0.1; b = torch.zeros(N, device=dev)
        y = torch.empty(M, N, device=dev)
        f = L.rnnt_mi355x_dense_forward
        f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        ACT = int(os.environ.get("ACT", "1"))
        for _ in range(3): assert f(p(x), p(w), p(b), M, K, N, ACT, p(y), None) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f(p(x), p(w), p(b), M, K, N, ACT, p(y), None)
        e1.record(); torch.cuda.synchronize()
        out.append(f"K={K} N={N}: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us")
    print(f"{name:10s}", " | ".join(out), flush=True)
