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
    assert _lib.lib().rnnt_mi355x_version() == 300
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
    assert f(1, 10, 1025, ctypes.byref(out)) == _lib.UNSUPPORTED_SIZE
    assert f(1 << 12, 1 << 10, 512, ctypes.byref(out)) == _lib.UNSUPPORTED_SIZE  # B*T*U1 >= 2^31
    assert f(1, 1 << 18, 1024, ctypes.byref(out)) == _lib.UNSUPPORTED_SIZE         # (maxT+maxU1+32)*maxU1 >= 2^27
    assert f(1, (1 << 17) - 1056, 1024, ctypes.byref(out)) == _lib.UNSUPPORTED_SIZE  # ... exactly 2^27
    assert f(1, (1 << 17) - 1057, 1024, ctypes.byref(out)) == 0                      # ... just below
    assert f(1, 10, 10, None) == _lib.INVALID_VALUE


def test_null_and_range_arguments_are_rejected_before_any_launch():
    L = _lib.lib()
    one = ctypes.c_void_p(256)  # never dereferenced: validation fails first
    assert L.rnnt_mi355x_forward(None, one, one, one, 1, 2, 2, 4, 0, one, one, None) == _lib.INVALID_VALUE
    assert L.rnnt_mi355x_forward(one, one, one, one, 1, 2, 2, 4, 4, one, one, None) == _lib.INVALID_VALUE
    assert L.rnnt_mi355x_forward(one, one, one, one, 1, 2, 2, 4, -1, one, one, None) == _lib.INVALID_VALUE
    nan = float("nan")
    assert L.rnnt_mi355x_backward(one, None, one, one, one, 1, 2, 2, 4, 0, None, one, None, nan) == _lib.INVALID_VALUE
    assert L.rnnt_mi355x_loss_grad(one, one, one, one, 1, 2, 1100, 4, 0, None, one, one, None, nan) == _lib.UNSUPPORTED_SIZE
    # a per-call threshold above -16 is refused like one set process-wide
    assert L.rnnt_mi355x_backward(one, one, one, one, one, 1, 2, 2, 4, 0, None, one, None, -8.0) == _lib.INVALID_VALUE
    assert L.rnnt_mi355x_reduce_scalars(None, 1, one, one, None) == _lib.INVALID_VALUE


def test_prune_threshold_setting_and_static_entry_point_validation():
    L = _lib.lib()
    one = ctypes.c_void_p(256)  # never dereferenced: validation fails first
    assert L.rnnt_mi355x_get_prune_log2() == _lib.DEFAULT_PRUNE_LOG2 == -28.0
    assert L.rnnt_mi355x_set_prune_log2(-8.0) == _lib.INVALID_VALUE       # would show at the 1e-4 bar
    assert L.rnnt_mi355x_set_prune_log2(float("nan")) == _lib.INVALID_VALUE
    assert L.rnnt_mi355x_set_prune_log2(-32.0) == _lib.OK
    assert L.rnnt_mi355x_get_prune_log2() == -32.0
    assert L.rnnt_mi355x_set_prune_log2(float("-inf")) == _lib.OK          # pruning off
    assert L.rnnt_mi355x_set_prune_log2(_lib.DEFAULT_PRUNE_LOG2) == _lib.OK
    # the entry point with the built-in workspace rejects what does not fit it, before touching a GPU
    assert _lib.workspace_bytes(8, 150, 41) < _lib.STATIC_WORKSPACE_BYTES < _lib.workspace_bytes(64, 400, 101)
    f = L.rnnt_mi355x_loss_grad_static
    nan = float("nan")
    assert f(one, one, one, one, 64, 400, 101, 1024, 0, None, one, None, None, None, nan) == _lib.UNSUPPORTED_SIZE
    assert f(one, one, one, one, 1, 2, 2, 4, 4, None, one, None, None, None, nan) == _lib.INVALID_VALUE   # blank outside [0,V)
    assert L.rnnt_mi355x_read_stats(None, None, None, None) == _lib.INVALID_VALUE


def test_python_layer_refuses_cpu_tensors():
    import torch
    import rnnt_amd
    acts = torch.zeros(1, 3, 2, 5)
    with pytest.raises(RuntimeError, match="no CPU path"):
        rnnt_amd.rnnt_loss(acts, torch.zeros(1, 1, dtype=torch.int32), torch.tensor([3]), torch.tensor([1]), 0)


def test_new_entry_points_validate_without_a_gpu():
    L = _lib.lib()
    one = ctypes.c_void_p(256)  # never dereferenced: validation fails first
    out = ctypes.c_size_t(0)
    # fused joint: the workspace depends on V and V is limited
    assert _lib.joint_workspace_bytes(4, 50, 11, 62) > _lib.workspace_bytes(4, 50, 11)
    assert _lib.joint_workspace_bytes(4, 50, 11, 1024) > _lib.joint_workspace_bytes(4, 50, 11, 62)
    assert L.rnnt_mi355x_joint_workspace_bytes(1, 4, 4, 4097, ctypes.byref(out)) == _lib.UNSUPPORTED_SIZE
    assert L.rnnt_mi355x_joint_forward(None, 0, one, one, one, one, 1, 2, 2, 4, 0, one, one, None) == _lib.INVALID_VALUE
    # crop as an argument
    assert L.rnnt_mi355x_forward_cropped(one, one, one, one, 1, 4, 4, 2, 4, 0, one, one, None) == _lib.INVALID_VALUE  # nothing left
    assert L.rnnt_mi355x_forward_cropped(one, one, one, one, 1, 4, -1, 2, 4, 0, one, one, None) == _lib.INVALID_VALUE
    assert L.rnnt_mi355x_backward_cropped(one, None, one, one, one, 1, 4, 2, 2, 4, 0, None, one, None,
                                          float("nan")) == _lib.INVALID_VALUE
    # FC-joint first layer and Dense layers
    assert _lib.fc_joint_workspace_bytes(2, 17, 5, 100) == 4 * 2 * 3 * 5 * 100   # ceil(17 / 8) frame chunks
    assert L.rnnt_mi355x_fc_joint_workspace_bytes(1, 1, 1, 1537, ctypes.byref(out)) == _lib.UNSUPPORTED_SIZE
    assert L.rnnt_mi355x_fc_joint_backward(one, one, one, 1, 1, 1, 4, one, one, None, None) == _lib.INVALID_VALUE
    assert _lib.dense_workspace_bytes(1000, 100, 62) > 0
    assert L.rnnt_mi355x_dense_workspace_bytes(10, 129, 4, ctypes.byref(out)) == _lib.UNSUPPORTED_SIZE
    assert L.rnnt_mi355x_dense_forward(one, one, None, 0, 4, 4, 0, one, None) == _lib.INVALID_VALUE
    assert L.rnnt_mi355x_dense_backward(one, one, None, one, 8, 4, 4, 1, one, one, one, one, None) == _lib.INVALID_VALUE  # tanh needs y
    # decoder step
    assert L.rnnt_mi355x_joint_log_softmax(one, 2, one, 2, None, None, 3, 8, one, one, None) == _lib.INVALID_VALUE  # 3 rows from 2
    assert L.rnnt_mi355x_joint_log_softmax(one, 2, None, 0, None, None, 2, 0, one, one, None) == _lib.INVALID_VALUE


def test_host_side_helpers_without_a_gpu():
    import numpy as np
    import torch
    import rnnt_amd
    from oracle import rnnt_oracle as orc
    yt, yp = torch.randn(2, 5, 3), torch.randn(2, 4, 2)
    np.testing.assert_array_equal(rnnt_amd.concatenate_lambda([yt, yp]).numpy(),
                                  orc.concatenate_lambda(yt.numpy(), yp.numpy()).astype(np.float32))
    with pytest.raises(ValueError, match="no CPU path"):
        rnnt_amd.fc_joint_first_layer(torch.zeros(1, 2, 4), torch.zeros(1, 3, 4))
    with pytest.raises(ValueError, match="no CPU path"):
        rnnt_amd.joint_log_softmax(torch.zeros(2, 4))
    with pytest.raises(ValueError, match="no CPU path"):
        rnnt_amd.joint_dense(torch.zeros(3, 4), torch.zeros(4, 2), torch.zeros(2))
    with pytest.raises(ValueError, match="no CPU path"):   # also beyond the widths of the hand-written kernels,
        rnnt_amd.joint_dense(torch.zeros(3, 200), torch.zeros(200, 300), torch.zeros(300))  # where a library GEMM runs
    with pytest.raises(ValueError):
        rnnt_amd.fc_joint(yt, yp, [(torch.zeros(7, 3), torch.zeros(3))])   # first kernel must have He + Hd rows
    assert rnnt_amd.softmax(torch.zeros(5), is_log=True, axis=0).tolist() == [1.0] * 5   # reference quirk, no GPU needed
