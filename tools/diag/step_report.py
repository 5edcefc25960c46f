#!/usr/bin/env python3
"""One training step out of a rocprofv3 --kernel-trace csv of tools/train_bench.py: the launches between the last two fused-Adam
kernels, their span, the busy union, the time per queue and per kernel family inside that step, and the launch sequence of the
main queue with the idle time in front of each launch.  python tools/diag/step_report.py <kernel_trace.csv> [--seq]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"].lower()]
assert len(adam) >= 2, "needs two optimizer steps in the trace"
# fused Adam is several launches per step: step boundary = a gap of more than 50 launches between two of them
bounds = [adam[0]] + [b for a, b in zip(adam, adam[1:]) if b - a > 50]
lo, hi = bounds[-2], bounds[-1]
step = rows[lo:hi]
t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
short = lambda n: re.sub(r"^void (gc::)?|^gc::|\(.*$", "", n)[:70]
by_q = collections.defaultdict(float)
by_k = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    by_q[r.get("Queue_Id", "?")] += d
    e = by_k[short(r["Kernel_Name"])]
    e[0] += 1
    e[1] += d
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
busy, cur = 0, ev[0][0]
for s, e in ev:
    if e > cur:
        busy += e - max(s, cur)
        cur = e
print(f"step: {len(step)} launches, span {(t1 - t0) / 1e6:.2f} ms, busy union {busy / 1e6:.2f} ms, sum of kernels {sum(by_q.values()) / 1e3:.2f} ms")
for q, us in sorted(by_q.items(), key=lambda kv: -kv[1]):
    print(f"  queue {q}: {us / 1e3:.2f} ms of kernels")
print("kernel families (launches, total us, avg us):")
for k, (n, us) in sorted(by_k.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"  {us:9.1f} us {n:5d} x {us / n:8.1f}  {k}")
if "--seq" in sys.argv:
    mainq = max(by_q, key=by_q.get)
    prev = t0
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        tag = "" if r.get("Queue_Id", "?") == mainq else "   [side]"
        print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  gap {max(0, s - prev) / 1e3:6.1f}  {short(r['Kernel_Name'])}{tag}")
        prev = max(prev, e)
