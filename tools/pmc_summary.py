#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel name.
    python tools/pmc_summary.py <dir> [regex]"""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
rx = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if rx and not rx.search(name):
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(acc.items()):
    print(name[:110])
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} n={len(v):4d} mean={sum(v) / len(v):16.1f}")
