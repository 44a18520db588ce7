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


def sources_digest() -> str:
    """Sixteen hex digits that change whenever a kernel source changes: measurements taken from a build (the HBM
    counter bytes of profiles/traffic.json) are stamped with it and not quoted for another one."""
    import hashlib
    h = hashlib.sha256()
    for p in sorted((*SOURCES, *PIECES)):
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


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
    at run time with RNNT_MI355X_LIB).

    Several processes may call this at once (ranks of one job, test workers): one of them builds, under a file
    lock, the others wait and find the library current; objects and the library are written under names of their
    own and moved into place, so nobody ever loads a half-written file."""
    import fcntl
    import tempfile
    if not force and not needs_build() and out == LIB:
        return LIB
    hipcc = hipcc_path()
    with open(out + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and out == LIB and not needs_build():  # another process built it while this one waited
                return LIB
            flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", *[f"-D{d}" for d in defines]]
            with tempfile.TemporaryDirectory(prefix=".build-", dir=os.path.dirname(out)) as tmp:
                # one object per source, compiled side by side, then one link
                objs = [os.path.join(tmp, os.path.splitext(os.path.basename(src))[0] + ".o") for src in SOURCES]
                cmds = [[hipcc, *flags, "-c", src, "-o", obj] for src, obj in zip(SOURCES, objs)]
                if verbose:
                    for cmd in cmds:
                        print(" ".join(cmd))
                procs = [subprocess.Popen(cmd) for cmd in cmds]
                failed = [cmd for cmd, pr in zip(cmds, procs) if pr.wait() != 0]
                if failed:
                    raise subprocess.CalledProcessError(1, failed[0])
                staged = os.path.join(tmp, os.path.basename(out))
                link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", staged, *objs]
                if verbose:
                    print(" ".join(link))
                subprocess.check_call(link)
                os.replace(staged, out)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
