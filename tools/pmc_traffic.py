#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
gfx950 corrections: both counters are in KiB; FETCH_SIZE reports half the bytes of
a wide coalesced read stream and is doubled; WRITE_SIZE is exact for 16-B-per-lane
streaming stores."""
import csv, glob, json, sys
from collections import defaultdict

# which step of the path a kernel belongs to (the gradient pass is several kernels: their bytes per launch add up)
GROUPS = (("k_lse", "k_lse"), ("k_skew_records", "k_sweep"), ("k_sweep", "k_sweep"), ("k_classify_rows", "k_grad"),
          ("k_fill_rows", "k_grad"), ("k_patch_sparse", "k_grad"), ("k_zero_frames", "k_grad"), ("k_grad", "k_grad"),
          ("k_init", "k_init"), ("k_reduce_scalars", "k_reduce_scalars"))


def per_kernel(dirpath, counter):
    by_name = defaultdict(list)
    for f in glob.glob(dirpath + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter: continue
            by_name[row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]].append(float(row["Counter_Value"]))
    acc = defaultdict(list)  # group -> one mean per kernel of the group
    for n, vals in by_name.items():
        for key, group in GROUPS:
            if key in n:
                acc[group].append((sum(vals) / len(vals), len(vals)))
                break
    return acc

fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f = sum(m for m, _ in fetch[k]) * 1024 * 2   # KiB -> B, x2 gfx950 correction
    w = sum(m for m, _ in write[k]) * 1024
    seen = max([n for _, n in fetch[k]] + [0])
    out[k] = {"launches_seen": seen, "kernels": len(fetch[k]), "read_bytes_per_launch": f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": f + w}
    print(f"{k:18s} kernels={len(fetch[k])} launches={seen:3d} read={f/1e9:8.3f} GB  write={w/1e9:8.3f} GB  total={(f+w)/1e9:8.3f} GB")
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
