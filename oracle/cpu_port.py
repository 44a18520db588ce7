"""ctypes loader for oracle/librnnt_oracle_cpu.so (the fp32 C restatement in
oracle/rnnt_cpu.c).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by rnnt_amd."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librnnt_oracle_cpu.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "rnnt_cpu.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "librnnt_oracle_cpu.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.rnnt_oracle_cpu_loss_grad.restype = ctypes.c_int
        _lib.rnnt_oracle_cpu_loss_grad.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3 + [ctypes.c_int]
        _lib.rnnt_oracle_cpu_max_threads.restype = ctypes.c_int
    return _lib


def max_threads() -> int:
    return int(lib().rnnt_oracle_cpu_max_threads())


def loss_grad(logits, labels, input_lengths, label_lengths, blank, grad_scale=None, want_grad=True, inplace=False, num_threads=0):
    """fp32 costs[B] (and gradient w.r.t. logits) from the C restatement.
    logits: float32 [B,T,U1,V]; with inplace=True the gradient overwrites it."""
    x = np.ascontiguousarray(logits, dtype=np.float32)
    B, T, U1, V = x.shape
    lab = np.ascontiguousarray(np.asarray(labels).reshape(B, -1), dtype=np.int32)
    if lab.shape[1] != U1 - 1:
        pad = np.zeros((B, U1 - 1), dtype=np.int32)
        pad[:, : min(U1 - 1, lab.shape[1])] = lab[:, : U1 - 1]
        lab = pad
    tl = np.ascontiguousarray(np.asarray(input_lengths).reshape(B), dtype=np.int32)
    ul = np.ascontiguousarray(np.asarray(label_lengths).reshape(B), dtype=np.int32)
    costs = np.zeros(B, dtype=np.float32)
    gs = None if grad_scale is None else np.ascontiguousarray(grad_scale, dtype=np.float32)
    grad = None
    if want_grad:
        grad = x if inplace else np.empty_like(x)
    rc = lib().rnnt_oracle_cpu_loss_grad(
        x.ctypes.data, lab.ctypes.data, tl.ctypes.data, ul.ctypes.data, B, T, U1, V, int(blank),
        None if gs is None else gs.ctypes.data, costs.ctypes.data, None if grad is None else grad.ctypes.data, int(num_threads))
    if rc != 0:
        raise RuntimeError(f"rnnt_oracle_cpu_loss_grad failed with status {rc}")
    return costs, grad
