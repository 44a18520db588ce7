"""Test helper: drives librnnt_mi355x.so through its C ABI (ctypes), using
torch only to own device memory.  Not a test module itself."""
import ctypes

import numpy as np
import torch

from rnnt_amd import _lib


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class HipRun:
    def __init__(self, logits, labels, in_len, lab_len, blank, grad_scale=None, device="cuda:0", prune_log2=None):
        self.dev = torch.device(device)
        self.prune = _lib.prune_arg(prune_log2)  # None = the process default
        x = np.ascontiguousarray(logits, dtype=np.float32)
        self.B, self.T, self.U1, self.V = x.shape
        self.blank = int(blank)
        self.x = torch.from_numpy(x).to(self.dev)
        lab = np.ascontiguousarray(np.asarray(labels, dtype=np.int32).reshape(self.B, -1))
        assert lab.shape[1] == self.U1 - 1
        self.lab = torch.from_numpy(lab).to(self.dev) if lab.size else torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.il = torch.from_numpy(np.ascontiguousarray(in_len, dtype=np.int32).reshape(self.B)).to(self.dev)
        self.ll = torch.from_numpy(np.ascontiguousarray(lab_len, dtype=np.int32).reshape(self.B)).to(self.dev)
        self.gs = None if grad_scale is None else torch.from_numpy(np.ascontiguousarray(grad_scale, dtype=np.float32)).to(self.dev)
        self.ws = torch.empty(_lib.workspace_bytes(self.B, self.T, self.U1), dtype=torch.uint8, device=self.dev)
        self.ws.fill_(0xFF)  # poison: every cell the kernels read must have been written by them (0xFF.. = NaN)
        self.costs = torch.full((self.B,), float("nan"), dtype=torch.float32, device=self.dev)
        self.stream = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def meta(self):
        return (self.B, self.T, self.U1, self.V, self.blank)

    def forward(self):
        rc = _lib.lib().rnnt_mi355x_forward(_p(self.x), _p(self.lab), _p(self.il), _p(self.ll), *self.meta(),
                                            _p(self.costs), _p(self.ws), self.stream)
        return rc

    def backward(self, out=None):
        g = self.x if out is None else out
        return _lib.lib().rnnt_mi355x_backward(_p(self.x), _p(g), _p(self.lab), _p(self.il), _p(self.ll), *self.meta(),
                                               _p(self.gs), _p(self.ws), self.stream, self.prune)

    def loss_grad(self):
        return _lib.lib().rnnt_mi355x_loss_grad(_p(self.x), _p(self.lab), _p(self.il), _p(self.ll), *self.meta(),
                                                _p(self.gs), _p(self.costs), _p(self.ws), self.stream, self.prune)

    def scalars(self):
        out = torch.empty(2, dtype=torch.float64, device=self.dev)
        rc = _lib.lib().rnnt_mi355x_reduce_scalars(_p(self.costs), self.B, _p(self.ws), _p(out), self.stream)
        assert rc == 0
        return out.cpu().numpy()

    def flags(self):
        f = ctypes.c_int32(0)
        rc = _lib.lib().rnnt_mi355x_read_status(_p(self.ws), ctypes.byref(f), self.stream)
        return rc, f.value


def make_batch(B, T, U, V, blank, seed, ragged=True, scale=1.0):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((B, T, U + 1, V)) * scale).astype(np.float32)
    lab = rng.integers(0, max(V - 1, 1), size=(B, U)).astype(np.int32)
    if V > 1:
        lab[lab >= blank] += 1
    if ragged and B > 1:
        il = rng.integers(max(1, T // 2), T + 1, size=B).astype(np.int32)
        ll = rng.integers(0, U + 1, size=B).astype(np.int32)
        il[0], ll[0] = T, U
    else:
        il, ll = np.full(B, T, np.int32), np.full(B, U, np.int32)
    return x, lab, il, ll
