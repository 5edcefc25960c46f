#!/usr/bin/env python3
"""Kernel statistics (the `--stats` table) from a rocprofv3 rocpd sqlite database, as CSV on stdout.
Usage: python tools/rocpd_stats.py results.db [--by-grid]   (by-grid: one row per (kernel, grid size))"""
import sqlite3, sys, statistics as st
db = sqlite3.connect(sys.argv[1])
by_grid = "--by-grid" in sys.argv
rows = db.execute("select name, duration, grid_x, grid_y, grid_z, workgroup_x from kernels").fetchall()
agg = {}
for name, dur, gx, gy, gz, wx in rows:
    key = (name, (gx // max(wx, 1), gy, gz)) if by_grid else (name, None)
    agg.setdefault(key, []).append(dur)
total = sum(sum(v) for v in agg.values())
print('"Name","Grid","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"')
for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f'"{name}","{grid if grid else ""}",{len(v)},{sum(v)},{sum(v)/len(v):.1f},{100*sum(v)/total:.2f},{min(v)},{max(v)},{st.pstdev(v):.1f}')
