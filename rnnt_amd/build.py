"""Compiles rnnt_amd/csrc/*.hip into librnnt_mi355x.so for gfx950.

hipcc cross-compiles without a GPU, so this runs in a CPU-only container; the
built library stays in-tree (git-ignored) and travels with the source tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = [os.path.join(CSRC, "rnnt_mi355x.hip"), os.path.join(CSRC, "fc_joint.hip"),
           os.path.join(CSRC, "skinny_dense.hip"), os.path.join(CSRC, "decode_step.hip")]
# rnnt_mi355x.hip is one translation unit whose kernels live in these included pieces
PIECES = [os.path.join(CSRC, n) for n in ("workspace.h", "device_util.h", "lse_rows.h", "sweep.h", "grad_rows.h", "grad_span.h", "grad_split.h", "cell_coef.h", "grad_flat.h", "lse_span.h",
                                          "small_vocab.h", "joint.h", "aux_kernels.h")]
HEADER = os.path.join(os.path.dirname(_HERE), "include", "rnnt_mi355x.h")
LIB = os.path.join(CSRC, "librnnt_mi355x.so")


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build librnnt_mi355x.so")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in (*SOURCES, *PIECES, HEADER))


def newer_sources():
    """Sources that are newer than the built library."""
    if not os.path.exists(LIB):
        return [*SOURCES]
    t = os.path.getmtime(LIB)
    return [p for p in (*SOURCES, *PIECES, HEADER) if os.path.getmtime(p) > t]


def build(force: bool = False, verbose: bool = False, defines=(), out: str = LIB) -> str:
    """defines / out: experiment builds (`-DNAME=VALUE` variants of the library under another file name, selected
    at run time with RNNT_MI355X_LIB)."""
    if not force and not needs_build() and out == LIB:
        return LIB
    hipcc = hipcc_path()
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", *[f"-D{d}" for d in defines]]
    # one object per source, compiled side by side (objects live next to the library, git-ignored), then one link
    tag = "" if out == LIB else "." + os.path.splitext(os.path.basename(out))[0]
    objs = [os.path.splitext(src)[0] + tag + ".o" for src in SOURCES]
    cmds = [[hipcc, *flags, "-c", src, "-o", obj] for src, obj in zip(SOURCES, objs)]
    if verbose:
        for cmd in cmds:
            print(" ".join(cmd))
    procs = [subprocess.Popen(cmd) for cmd in cmds]
    failed = [cmd for cmd, pr in zip(cmds, procs) if pr.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    for obj in objs:
        os.remove(obj)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
