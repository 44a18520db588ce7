#!/usr/bin/env python3
"""Weight-gradient time of the Dense layer (K, N pairs; no activation) through the C ABI, for each library variant
given (tools/variants/<name>.so), e.g. the knock-out builds RNNT_WGRAD_NO_LOAD / RNNT_WGRAD_NO_MFMA."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
M = 32 * 500 * 101
p = lambda t: ctypes.c_void_p(t.data_ptr())
for name in sys.argv[1:]:
    L = ctypes.CDLL(os.path.join(ROOT, "tools", "variants", name + ".so"))
    out = []
    for K, N in ((100, 62), (100, 100), (128, 128), (64, 40)):
        x = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev) * 0.1
        dy = torch.randn(M, N, device=dev); dw = torch.empty(K, N, device=dev); db = torch.empty(N, device=dev)
        nbytes = ctypes.c_size_t(0)
        L.rnnt_mi355x_dense_workspace_bytes.argtypes = [ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]
        assert L.rnnt_mi355x_dense_workspace_bytes(M, K, N, ctypes.byref(nbytes)) == 0
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        f = L.rnnt_mi355x_dense_backward
        f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5
        call = lambda: f(p(x), p(w), None, p(dy), M, K, N, 0, None, p(dw), p(db), p(ws), None)
        for _ in range(3): assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        out.append(f"K={K} N={N}: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us")
    print(f"{name:10s}", " | ".join(out), flush=True)
