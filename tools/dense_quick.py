#!/usr/bin/env python3
"""Dense layer over the joint lattice: forward and backward times for a few (K, N) at M = 32*500*101."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rnnt_amd
dev = torch.device("cuda:0")
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
M = 32 * 500 * 101
ZERO_FROM = int(sys.argv[1]) if len(sys.argv) > 1 else 101   # label positions u >= this get a zero upstream gradient
for K, N in ((100, 62), (100, 100), (128, 128), (64, 40), (32, 32)):
    x = torch.randn(M, K, device=dev, requires_grad=True)
    k = (torch.randn(K, N, device=dev) * 0.1).requires_grad_(True); b = torch.zeros(N, device=dev, requires_grad=True)
    dy = torch.randn(M, N, device=dev)
    dy.view(32, 500, 101, N)[:, :, ZERO_FROM:, :] = 0
    fwd = timeit(lambda: rnnt_amd.joint_dense(x, k, b, "tanh"))
    def fb():
        x.grad = k.grad = b.grad = None
        rnnt_amd.joint_dense(x, k, b, "tanh").backward(dy)
    both = timeit(fb)
    print(f"zero rows {1 - min(ZERO_FROM, 101) / 101:.0%} dense M={M} K={K:4d} N={N:4d}: forward {fwd * 1e3:8.1f} us ({M * (K + N) * 4 / fwd / 1e9:5.2f} TB/s)  backward {(both - fwd) * 1e3:8.1f} us ({M * (2 * K + 2 * N) * 4 / (both - fwd) / 1e9:5.2f} TB/s)", flush=True)
