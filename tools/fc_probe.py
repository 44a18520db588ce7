#!/usr/bin/env python3
"""FC-joint building blocks over their size ranges: fc_joint_first_layer (tanh(pe + pd) and its backward) over H,
joint_dense over (K, N); time and effective bandwidth of forward and backward."""
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

B, T, U1 = 32, 500, 101
for H in (16, 62, 64, 100, 128, 200, 256, 512, 1024, 1536):
    pe = torch.randn(B, T, H, device=dev, requires_grad=True); pd = torch.randn(B, U1, H, device=dev, requires_grad=True)
    h = rnnt_amd.fc_joint_first_layer(pe, pd)
    dh = torch.randn_like(h)
    fwd = timeit(lambda: rnnt_amd.fc_joint_first_layer(pe, pd))
    def fb():
        pe.grad = pd.grad = None
        rnnt_amd.fc_joint_first_layer(pe, pd).backward(dh)
    both = timeit(fb)
    nb = B * T * U1 * H * 4
    print(f"first layer H={H:5d}: forward {fwd * 1e3:8.1f} us ({nb / fwd / 1e9:5.2f} TB/s written)  backward {(both - fwd) * 1e3:8.1f} us ({nb / (both - fwd) / 1e9:5.2f} TB/s read)", flush=True)
    del h, dh
M = B * T * U1
for K, N in ((100, 62), (100, 100), (128, 128), (64, 40), (32, 32), (16, 8), (128, 62), (100, 128), (17, 33)):
    x = torch.randn(M, K, device=dev, requires_grad=True)
    k = (torch.randn(K, N, device=dev) * 0.1).requires_grad_(True); b = torch.zeros(N, device=dev, requires_grad=True)
    dy = torch.randn(M, N, device=dev)
    fwd = timeit(lambda: rnnt_amd.joint_dense(x, k, b, "tanh"))
    def fb():
        x.grad = k.grad = b.grad = None
        rnnt_amd.joint_dense(x, k, b, "tanh").backward(dy)
    both = timeit(fb)
    print(f"dense M={M} K={K:4d} N={N:4d}: forward {fwd * 1e3:8.1f} us ({M * (K + N) * 4 / fwd / 1e9:5.2f} TB/s)  backward {(both - fwd) * 1e3:8.1f} us ({M * (2 * K + 2 * N) * 4 / (both - fwd) / 1e9:5.2f} TB/s)", flush=True)
