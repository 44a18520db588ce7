#!/usr/bin/env python3
"""Prints a compact per-kernel table from a rocprofv3 *_kernel_stats.csv."""
import csv, glob, sys
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_stats.csv", recursive=True) if not path.endswith(".csv") else [path]
for f in files:
    for row in csv.DictReader(open(f)):
        n = row["Name"].replace("(anonymous namespace)::", "")
        if n.startswith("void at::") : continue
        print(f'{n[:48]:50s} calls={row["Calls"]:>4s} avg_us={float(row["AverageNs"])/1e3:10.1f} min_us={float(row["MinNs"])/1e3:10.1f} pct={row["Percentage"]}')
