#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
gfx950 corrections: both counters are in KiB; FETCH_SIZE reports half the bytes of
a wide coalesced read stream and is doubled; WRITE_SIZE is exact for 16-B-per-lane
streaming stores."""
import csv, glob, json, sys
from collections import defaultdict

def per_kernel(dirpath, counter):
    acc = defaultdict(list)
    for f in glob.glob(dirpath + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter: continue
            n = row["Kernel_Name"].replace("(anonymous namespace)::", "")
            for key in ("k_grad", "k_lse", "k_sweep", "k_init", "k_reduce_scalars"):
                if key in n.split("(")[0]:
                    acc[key].append(float(row["Counter_Value"]))
    return acc

fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f = sum(fetch[k]) / max(len(fetch[k]), 1) * 1024 * 2   # KiB -> B, x2 gfx950 correction
    w = sum(write[k]) / max(len(write[k]), 1) * 1024
    out[k] = {"launches_seen": len(fetch[k]), "read_bytes_per_launch": f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": f + w}
    print(f"{k:18s} launches={len(fetch[k]):3d} read={f/1e9:8.3f} GB  write={w/1e9:8.3f} GB  total={(f+w)/1e9:8.3f} GB")
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
