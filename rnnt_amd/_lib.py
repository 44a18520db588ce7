"""ctypes binding of librnnt_mi355x.so (C ABI: include/rnnt_mi355x.h).

There is no fallback: if the HIP library is missing or fails to load, every
use of the loss raises.
"""
from __future__ import annotations

import ctypes
import subprocess
import os

from . import build as _build
from .build import LIB as _DEFAULT_LIB

# RNNT_MI355X_LIB selects another build of the same library (debug / experiment builds).
LIB = os.environ.get("RNNT_MI355X_LIB") or _DEFAULT_LIB

OK = 0
INVALID_VALUE = 1
UNSUPPORTED_SIZE = 2
LAUNCH_FAILURE = 3
BAD_INPUT_DATA = 4
FLAG_BAD_LENGTH = 1
FLAG_BAD_LABEL = 2
DEFAULT_PRUNE_LOG2 = -28.0  # kPruneLog2 of csrc/workspace.h
USE_DEFAULT_PRUNE = float("nan")  # prune_log2 argument meaning "the process default of rnnt_mi355x_set_prune_log2"


def prune_arg(prune_log2) -> float:
    """The prune_log2 argument of the C entry points for a Python-side value
    (None = the process default)."""
    if prune_log2 is None:
        return USE_DEFAULT_PRUNE
    v = float(prune_log2)
    if not v <= -16.0:  # also refuses NaN: the default is spelled None here
        raise ValueError("prune_log2 must lie in [-inf, -16]")
    return v
STATIC_WORKSPACE_BYTES = 32 << 20  # RNNT_MI355X_STATIC_WORKSPACE_BYTES

_lib = None

_vp, _i, _sz, _ll, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_longlong, ctypes.c_float
_META = [_i] * 5  # B, maxT, maxU1, V, blank

_SIGNATURES = {
    "rnnt_mi355x_workspace_bytes": (ctypes.c_int, [_i, _i, _i, ctypes.POINTER(_sz)]),
    "rnnt_mi355x_forward": (ctypes.c_int, [_vp] * 4 + _META + [_vp, _vp, _vp]),
    "rnnt_mi355x_backward": (ctypes.c_int, [_vp] * 5 + _META + [_vp, _vp, _vp, _f]),
    "rnnt_mi355x_forward_strided": (ctypes.c_int, [_vp, _sz] + [_vp] * 3 + _META + [_vp, _vp, _vp]),
    "rnnt_mi355x_backward_strided": (ctypes.c_int, [_vp, _sz, _vp, _sz] + [_vp] * 3 + _META + [_vp, _vp, _vp, _f]),
    "rnnt_mi355x_forward_cropped": (ctypes.c_int, [_vp] * 4 + [_i] * 6 + [_vp, _vp, _vp]),
    "rnnt_mi355x_backward_cropped": (ctypes.c_int, [_vp] * 5 + [_i] * 6 + [_vp, _vp, _vp, _f]),
    "rnnt_mi355x_backward_inplace": (ctypes.c_int, [_vp] * 4 + _META + [_vp, _vp, _vp, _f]),
    "rnnt_mi355x_loss_grad": (ctypes.c_int, [_vp] * 4 + _META + [_vp, _vp, _vp, _vp, _f]),
    "rnnt_mi355x_loss_grad_events": (ctypes.c_int, [_vp] * 4 + _META + [_vp, _vp, _vp, _vp, _f, ctypes.POINTER(_vp)]),
    "rnnt_mi355x_loss_grad_static": (ctypes.c_int, [_vp] * 4 + _META + [_vp, _vp, _vp, _vp, _vp, _f]),
    "rnnt_mi355x_set_prune_log2": (ctypes.c_int, [ctypes.c_float]),
    "rnnt_mi355x_get_prune_log2": (ctypes.c_float, []),
    "rnnt_mi355x_read_stats": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), _vp]),
    "rnnt_mi355x_joint_workspace_bytes": (ctypes.c_int, [_i, _i, _i, _i, ctypes.POINTER(_sz)]),
    "rnnt_mi355x_joint_forward": (ctypes.c_int, [_vp, _sz] + [_vp] * 4 + _META + [_vp, _vp, _vp]),
    "rnnt_mi355x_joint_backward": (ctypes.c_int, [_vp, _sz] + [_vp] * 4 + _META + [_vp, _vp, _sz, _vp, _vp, _vp, _f]),
    "rnnt_mi355x_fc_joint_workspace_bytes": (ctypes.c_int, [_i, _i, _i, _i, ctypes.POINTER(_sz)]),
    "rnnt_mi355x_fc_joint_forward": (ctypes.c_int, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "rnnt_mi355x_fc_joint_backward": (ctypes.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "rnnt_mi355x_dense_workspace_bytes": (ctypes.c_int, [_ll, _i, _i, ctypes.POINTER(_sz)]),
    "rnnt_mi355x_dense_forward": (ctypes.c_int, [_vp, _vp, _vp, _ll, _i, _i, _i, _vp, _vp]),
    "rnnt_mi355x_dense_backward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _ll, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "rnnt_mi355x_joint_log_softmax": (ctypes.c_int, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "rnnt_mi355x_reduce_scalars": (ctypes.c_int, [_vp, _i, _vp, _vp, _vp]),
    "rnnt_mi355x_read_status": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int32), _vp]),
    "rnnt_mi355x_status_string": (ctypes.c_char_p, [_i]),
    "rnnt_mi355x_version": (ctypes.c_int, []),
}


class RNNTLibraryError(RuntimeError):
    pass


def exported_symbols():
    """Names every entry point include/rnnt_mi355x.h declares."""
    return tuple(_SIGNATURES)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RNNTLibraryError(
                f"{LIB} is missing: build it with `python -m rnnt_amd.build` (needs hipcc); "
                "rnnt_amd has no CPU or eager fallback")
        if LIB == _DEFAULT_LIB and _build.needs_build():
            # a source is newer than the binary: rebuild where a compiler is at hand, never run the stale one silently
            try:
                _build.hipcc_path()
            except RuntimeError:
                raise RNNTLibraryError(f"{LIB} is older than its sources ({_build.newer_sources()}) and hipcc is not "
                                       "available to rebuild it") from None
            try:
                _build.build()
            except (OSError, subprocess.CalledProcessError) as e:
                raise RNNTLibraryError(f"{LIB} is older than its sources and rebuilding it failed: {e}") from e
        try:
            handle = ctypes.CDLL(LIB)
        except OSError as e:
            raise RNNTLibraryError(f"cannot load {LIB}: {e}") from e
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.rnnt_mi355x_version() < 0 and os.environ.get("RNNT_MI355X_ALLOW_KNOCKOUT") != "1":
            raise RNNTLibraryError(f"{LIB} is a timing-experiment build with a piece of a kernel knocked out (its "
                                   "results are wrong); set RNNT_MI355X_ALLOW_KNOCKOUT=1 to time it")
        _lib = handle
    return _lib


def status_string(rc: int) -> str:
    return lib().rnnt_mi355x_status_string(int(rc)).decode()


def check(rc: int, what: str) -> None:
    if rc != OK:
        raise RuntimeError(f"{what}: {status_string(rc)} (status {rc})")


def fc_joint_workspace_bytes(B: int, T: int, U1: int, H: int) -> int:
    out = _sz(0)
    check(lib().rnnt_mi355x_fc_joint_workspace_bytes(B, T, U1, H, ctypes.byref(out)), "rnnt_mi355x_fc_joint_workspace_bytes")
    return int(out.value)


def dense_workspace_bytes(M: int, K: int, N: int) -> int:
    out = _sz(0)
    check(lib().rnnt_mi355x_dense_workspace_bytes(M, K, N, ctypes.byref(out)), "rnnt_mi355x_dense_workspace_bytes")
    return int(out.value)


def joint_workspace_bytes(B: int, maxT: int, maxU1: int, V: int) -> int:
    out = _sz(0)
    check(lib().rnnt_mi355x_joint_workspace_bytes(B, maxT, maxU1, V, ctypes.byref(out)), "rnnt_mi355x_joint_workspace_bytes")
    return int(out.value)


def read_stats(workspace_ptr, stream_ptr):
    """(lattice cells, pruned cells) of the last backward on that workspace."""
    cells, pruned = ctypes.c_uint64(0), ctypes.c_uint64(0)
    check(lib().rnnt_mi355x_read_stats(workspace_ptr, ctypes.byref(cells), ctypes.byref(pruned), stream_ptr),
          "rnnt_mi355x_read_stats")
    return int(cells.value), int(pruned.value)


def workspace_bytes(B: int, maxT: int, maxU1: int) -> int:
    out = _sz(0)
    check(lib().rnnt_mi355x_workspace_bytes(B, maxT, maxU1, ctypes.byref(out)), "rnnt_mi355x_workspace_bytes")
    return int(out.value)
