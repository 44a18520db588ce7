#!/usr/bin/env python3
"""Folds gpurun_out/prof_c<N>/traffic_kernels.json (tools/profile_hbm.sh) into profiles/traffic.json, the
per-kernel HBM counter bytes bench.py quotes.  usage: fold_traffic.py <config> <B> <prune_log2>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rnnt_amd import build  # noqa: E402
cfg, B, thr = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
k = json.load(open(os.path.join(ROOT, "gpurun_out", f"prof_c{cfg}", "traffic_kernels.json")))
path = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(path))
t["_note"] = ("HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes of bench.py --steps 2 "
              "--warmup 1 --ring 3, tools/profile_hbm.sh). KiB->B; FETCH_SIZE doubled (gfx950 reports half of a wide "
              "coalesced read stream); WRITE_SIZE exact for 16-B streaming stores.")
# stamped with the kernel sources it was measured on: bench.py quotes it for that build only
ent = {"B": B, "prune_log2": thr, "csrc_sha16": build.sources_digest()}
for name in ("k_lse", "k_sweep", "k_grad"):
    ent[f"{name}_hbm_bytes_per_launch"] = k[name]["hbm_bytes_per_launch"]
    ent[f"{name}_read"] = k[name]["read_bytes_per_launch"]
    ent[f"{name}_write"] = k[name]["write_bytes_per_launch"]
ent["step_hbm_bytes"] = sum(v["hbm_bytes_per_launch"] for v in k.values())
t[f"config{cfg}"] = ent
json.dump(t, open(path, "w"), indent=1)
print(json.dumps(ent, indent=1))
