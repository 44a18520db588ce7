#!/usr/bin/env python3
"""Condenses bench.py JSON lines read from stdin."""
import json, sys
for l in sys.stdin:
    l = l.strip()
    if l.startswith("{"):
        d = json.loads(l)
        print(sys.argv[1] if len(sys.argv) > 1 else "", "gpus", d["n_gpus"], "ms", round(d["ms_per_step"], 3), "Mcells/s", round(d["value"] / 1e6, 1),
              "step_frac", round(d["step_hbm_frac_of_peak"], 3), "kgrad_ms", round(d["roofline"]["kernel_ms"], 3), "kgrad_frac", round(d["roofline"]["frac"], 3))
