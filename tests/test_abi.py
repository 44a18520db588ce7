"""The C-ABI shared library loads on a CPU-only host and exports every symbol
include/rnnt_mi355x.h declares; argument validation that needs no GPU."""
import ctypes
import os
import re

import pytest

from rnnt_amd import _lib
from rnnt_amd.build import HEADER, LIB


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rnnt_mi355x_[a-z_0-9]+)\s*\(", text)))


def test_library_is_built_in_tree():
    assert os.path.exists(LIB), "run `python -m rnnt_amd.build` (hipcc cross-compiles without a GPU)"


def test_every_declared_symbol_is_exported():
    declared = _declared_symbols()
    assert len(declared) >= 9
    handle = ctypes.CDLL(LIB)
    for name in declared:
        assert hasattr(handle, name), name
    assert sorted(_lib.exported_symbols()) == declared


def test_version_and_status_strings():
    assert _lib.lib().rnnt_mi355x_version() == 101
    assert _lib.status_string(0) == "ok"
    for rc in (1, 2, 3, 4, 99):
        assert _lib.status_string(rc)


def test_workspace_query_and_shape_validation():
    n = _lib.workspace_bytes(64, 400, 101)
    assert 50e6 < n < 200e6
    assert _lib.workspace_bytes(1, 1, 1) >= 256
    out = ctypes.c_size_t(0)
    f = _lib.lib().rnnt_mi355x_workspace_bytes
    assert f(0, 10, 10, ctypes.byref(out)) == _lib.INVALID_VALUE
    assert f(1, 10, 513, ctypes.byref(out)) == _lib.UNSUPPORTED_SIZE
    assert f(1 << 12, 1 << 10, 512, ctypes.byref(out)) == _lib.UNSUPPORTED_SIZE  # B*T*U1 >= 2^31
    assert f(1, 10, 10, None) == _lib.INVALID_VALUE


def test_null_and_range_arguments_are_rejected_before_any_launch():
    L = _lib.lib()
    one = ctypes.c_void_p(256)  # never dereferenced: validation fails first
    assert L.rnnt_mi355x_forward(None, one, one, one, 1, 2, 2, 4, 0, one, one, None) == _lib.INVALID_VALUE
    assert L.rnnt_mi355x_forward(one, one, one, one, 1, 2, 2, 4, 4, one, one, None) == _lib.INVALID_VALUE
    assert L.rnnt_mi355x_forward(one, one, one, one, 1, 2, 2, 4, -1, one, one, None) == _lib.INVALID_VALUE
    assert L.rnnt_mi355x_backward(one, None, one, one, one, 1, 2, 2, 4, 0, None, one, None) == _lib.INVALID_VALUE
    assert L.rnnt_mi355x_loss_grad(one, one, one, one, 1, 2, 600, 4, 0, None, one, one, None) == _lib.UNSUPPORTED_SIZE
    assert L.rnnt_mi355x_reduce_scalars(None, 1, one, one, None) == _lib.INVALID_VALUE


def test_python_layer_refuses_cpu_tensors():
    import torch
    import rnnt_amd
    acts = torch.zeros(1, 3, 2, 5)
    with pytest.raises(RuntimeError, match="no CPU path"):
        rnnt_amd.rnnt_loss(acts, torch.zeros(1, 1, dtype=torch.int32), torch.tensor([3]), torch.tensor([1]), 0)
