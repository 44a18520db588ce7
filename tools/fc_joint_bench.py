#!/usr/bin/env python3
"""Times the FC-joint loss head (rnnt_amd.fc_joint_loss: split first layer + HIP broadcast-tanh + loss) against the
literal composition the reference builds (tiled concat -> Dense stack -> crop -> loss), forward + backward, on one GPU.
Shape: the reference's sample FC-joint model (enc 2x128, dec 128, joint 100 x 2 layers, V = 62) at B=32 T=500 U=100."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rnnt_amd
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32); ap.add_argument("--T", type=int, default=502); ap.add_argument("--U", type=int, default=100)
ap.add_argument("--He", type=int, default=256); ap.add_argument("--Hd", type=int, default=128); ap.add_argument("--H", type=int, default=100)
ap.add_argument("--V", type=int, default=62); ap.add_argument("--layers", type=int, default=2); ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--ours-only", action="store_true", help="skip the literal composition (for a kernel profile of the fused head alone)")
a = ap.parse_args()
dev = torch.device("cuda:0"); B, T0, U, V = a.B, a.T, a.U, a.V
yt = torch.randn(B, T0, a.He, device=dev, requires_grad=True); yp = torch.randn(B, U + 1, a.Hd, device=dev, requires_grad=True)
dims = [a.He + a.Hd] + [a.H] * (a.layers - 1) + [V]
ws = [(torch.empty(dims[i], dims[i + 1], device=dev).uniform_(-0.1, 0.1).requires_grad_(True),
       torch.empty(dims[i + 1], device=dev).uniform_(-0.1, 0.1).requires_grad_(True)) for i in range(a.layers)]
lab = torch.randint(0, 39, (B, U), device=dev, dtype=torch.int32)
il = torch.full((B, 1), T0, dtype=torch.int32, device=dev); ll = torch.full((B, 1), U, dtype=torch.int32, device=dev)
params = [yt, yp] + [p for kb in ws for p in kb]
def ours():
    for p in params: p.grad = None
    rnnt_amd.fc_joint_loss(yt, yp, ws, lab, il, ll).mean().backward()
def literal():
    for p in params: p.grad = None
    x = rnnt_amd.concatenate_lambda([yt, yp])
    for i, (k, b) in enumerate(ws):
        x = x @ k + b
        if i < len(ws) - 1: x = torch.tanh(x)
    rnnt_amd.rnnt_lambda_func_v2([x, lab, il, ll]).mean().backward()
def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps
ours(); g1 = [p.grad.clone() for p in params]
if a.ours_only:
    print(f"ours {timeit(ours):.3f} ms")
    sys.exit(0)
literal(); g2 = [p.grad.clone() for p in params]
print("max grad difference ours vs literal:", max((x - y).abs().max().item() for x, y in zip(g1, g2)))
t1, t2 = timeit(ours), timeit(literal)
print(f"FC joint head B={B} T0={T0} U={U} He={a.He} Hd={a.Hd} H={a.H} V={V} layers={a.layers}: ours {t1:.3f} ms | literal (tiled concat) {t2:.3f} ms | speed-up {t2/t1:.2f}x")
