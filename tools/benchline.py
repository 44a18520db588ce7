#!/usr/bin/env python3
"""Condenses bench.py JSON lines: one short line per file given (or per line of stdin)."""
import json, sys


def short(tag, d):
    k = {e["name"]: round(e["ms"], 3) for e in d["roofline"].get("kernels", [])}
    legs = {n: round(v["ms_per_step"], 3) for n, v in (d.get("legs") or {}).items()}
    print(tag, "gpus", d["n_gpus"], "ms", round(d["ms_per_step"], 4), "Mcells/s", round(d["value"] / 1e6, 1), "frac",
          round(d["roofline"]["frac"], 4), k, "pruned", round(d.get("pruned_row_frac") or 0, 3), legs or "")


if len(sys.argv) > 1:
    for f in sys.argv[1:]:
        short(f, json.loads(open(f).read().strip().splitlines()[-1]))
else:
    for l in sys.stdin:
        if l.strip().startswith("{"):
            short("", json.loads(l))
