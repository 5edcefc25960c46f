#!/usr/bin/env python3
"""Per-(kernel, grid) statistics from a rocprofv3 kernel_trace.csv (--kernel-trace --output-format csv), as CSV on stdout.
Usage: python tools/trace_by_grid.py <kernel_trace.csv> [name-substring ...]"""
import csv, statistics as st, sys
rows = csv.DictReader(open(sys.argv[1]))
pats = sys.argv[2:]
agg = {}
for r in rows:
    name = r["Kernel_Name"]
    if pats and not any(p in name for p in pats):
        continue
    wg = max(int(r["Workgroup_Size_X"]), 1)
    key = (name, int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    agg.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
total = sum(sum(v) for v in agg.values())
print('"Name","Grid","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for (name, gx, gy, gz), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f'"{name[:100]}","{gx}x{gy}x{gz}",{len(v)},{sum(v)},{sum(v)/len(v):.1f},{100*sum(v)/total:.2f},{min(v)},{max(v)}')
