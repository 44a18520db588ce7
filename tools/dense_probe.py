#!/usr/bin/env python3
"""Runs the tall-skinny Dense forward / backward a few times at the FC-joint sample shape (for rocprofv3 passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rnnt_amd
M, K, N = 32 * 500 * 101, 100, 62
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev, requires_grad=True)
k = torch.empty(K, N, device=dev).uniform_(-0.1, 0.1).requires_grad_(True)
b = torch.zeros(N, device=dev, requires_grad=True)
dy = torch.randn(M, N, device=dev)
for _ in range(4):
    y = rnnt_amd.joint_dense(x, k, b, "linear")
    y.backward(dy)
    x.grad = k.grad = b.grad = None
torch.cuda.synchronize()
print("done")
