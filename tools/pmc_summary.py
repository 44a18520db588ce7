#!/usr/bin/env python3
"""The library's kernels out of a rocprofv3 --pmc pass (counter_collection.csv under <dir>) as a small table:
dispatch, kernel (shortened), counter, value.  usage: pmc_summary.py <dir> <out.csv>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        if name.startswith("void "):
            head = name[5:]
        else:
            head = name
        if not head.startswith("k_"):
            continue
        rows.append((int(r["Dispatch_Id"]), name[:60], r["Counter_Name"], float(r["Counter_Value"])))
rows.sort()
with open(sys.argv[2], "w", newline="") as out:
    w = csv.writer(out)
    w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
    for d, n, c, v in rows:
        w.writerow([d, n, c, f"{v:.6f}"])
print(len(rows), "rows ->", sys.argv[2])
