#!/usr/bin/env python3
"""Prints the kernel timeline (start/end relative, us) of the last N dispatches of our kernels from a rocprofv3 kernel trace."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
rows = [r for r in csv.DictReader(open(f)) if "anonymous namespace" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:28]
    print(f'{name:30s} q={r.get("Queue_Id","?"):>3s} grid={r.get("Grid_Size_X", r.get("Grid_Size", "?")):>9s} start={(int(r["Start_Timestamp"])-t0)/1e3:9.1f} end={(int(r["End_Timestamp"])-t0)/1e3:9.1f} dur={(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f}')
