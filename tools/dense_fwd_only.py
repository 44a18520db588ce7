#!/usr/bin/env python3
"""A few launches of the Dense forward kernel (K=100, N=62 by default) for profiler passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rnnt_amd
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
N = int(sys.argv[2]) if len(sys.argv) > 2 else 62
dev = torch.device("cuda:0")
M = 32 * 500 * 101
x = torch.randn(M, K, device=dev); k = torch.randn(K, N, device=dev) * 0.1; b = torch.zeros(N, device=dev)
for _ in range(4):
    y = rnnt_amd.joint_dense(x, k, b, "tanh")
torch.cuda.synchronize()
