#!/usr/bin/env python3
"""Builds experiment variants of librnnt_mi355x.so (compile-time knobs of the
streaming kernels) into tools/variants/<name>.so; a run selects one with
RNNT_MI355X_LIB=<path>.  usage: build_variants.py name:DEF1=V,DEF2=V ..."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnnt_amd import build  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants")


def one(spec):
    name, _, defs = spec.partition(":")
    defines = [d for d in defs.split(",") if d]
    path = os.path.join(OUT, f"{name}.so")
    build.build(force=True, defines=defines, out=path)
    return path


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    with ThreadPoolExecutor(2) as ex:
        for p in ex.map(one, sys.argv[1:]):
            print(p)
