#!/usr/bin/env python3
"""Runs bench.py once per library variant in tools/variants (RNNT_MI355X_LIB) and
per input kind, and prints one line per run: step ms and the three kernel times.
usage: variant_bench.py [--config N] [--extra "bench args"] name ..."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--extra", default="")
ap.add_argument("--kinds", default="n1,peaked")
ap.add_argument("names", nargs="+")
a = ap.parse_args()
kinds = {"n1": [], "peaked": ["--logit-scale", "10"], "off": ["--prune-log2=-inf"], "p28": ["--prune-log2=-28"]}
for name in a.names:
    lib = os.path.join(ROOT, "tools", "variants", name + ".so")
    for kind in a.kinds.split(","):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(a.config), "--no-cpu-baseline", "--no-legs",
               "--steps", "12", "--warmup", "2", *kinds[kind], *a.extra.split()]
        r = subprocess.run(cmd, env=dict(os.environ, RNNT_MI355X_LIB=lib, RNNT_MI355X_ALLOW_KNOCKOUT="1"), capture_output=True, text=True, timeout=280)
        if r.returncode != 0:
            print(name, kind, "FAILED", r.stderr[-300:], flush=True)
            sys.exit(1)
        j = json.loads(r.stdout.strip().splitlines()[-1])
        k = {e["name"]: e["ms"] for e in j["roofline"]["kernels"]}
        print(f"{name:10s} {kind:7s} step {j['ms_per_step']:.3f} ms  dev-median {j['step_ms_device']['median']:.3f}  lse {k['k_lse']:.3f}  "
              f"sweep {k['k_sweep']:.3f}  grad {k['k_grad']:.3f}  pruned {j['pruned_row_frac']:.3f}  frac {j['roofline']['frac']:.3f}", flush=True)
