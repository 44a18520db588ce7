#!/usr/bin/env python3
"""joint_log_softmax (the decoders' inner step, batched) over row counts and vocabularies: time and bytes moved
(two rows read, one written per output row)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rnnt_amd
dev = torch.device("cuda:0")
for n, V in ((16, 62), (1024, 62), (200000, 62), (200000, 40), (200000, 128), (100000, 1024), (20000, 4096), (20000, 5000)):
    enc = torch.randn(n, V, device=dev); pred = torch.randn(n, V, device=dev)
    fn = lambda: rnnt_amd.joint_log_softmax(enc, pred)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"rows={n:7d} V={V:5d}  {ms * 1e3:9.1f} us  {3 * n * V * 4 / ms / 1e9:6.2f} TB/s", flush=True)
