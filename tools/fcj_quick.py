#!/usr/bin/env python3
"""First layer of the FC joint (tanh(pe + pd)) forward / backward times over H, through the torch wrapper."""
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
out = []
for H in ([int(a) for a in sys.argv[1:]] or (16, 64, 100, 128, 200, 256, 512, 1024)):
    pe = torch.randn(B, T, H, device=dev, requires_grad=True); pd = torch.randn(B, U1, H, device=dev, requires_grad=True)
    h = rnnt_amd.fc_joint_first_layer(pe, pd)
    dh = torch.randn_like(h)
    fwd = timeit(lambda: rnnt_amd.fc_joint_first_layer(pe, pd))
    def fb():
        pe.grad = pd.grad = None
        rnnt_amd.fc_joint_first_layer(pe, pd).backward(dh)
    both = timeit(fb)
    nb = B * T * U1 * H * 4
    out.append(f"H={H}: bwd {(both - fwd) * 1e3:.0f} us {nb / (both - fwd) / 1e9:.2f} TB/s")
    del h, dh
print(os.environ.get("RNNT_MI355X_LIB", "default").split("/")[-1], " | ".join(out), flush=True)
