#!/usr/bin/env python3
"""GPU busy / idle split of a rocprofv3 --kernel-trace csv: union of kernel intervals over launches 40 % .. 95 % of the trace (by count), and the kernels
that most often END right before an idle gap (the host was late with the next launch).  python tools/diag/gap_report.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ev = ev[int(0.4 * len(ev)):int(0.95 * len(ev))]   # by launch count: the steady-state steps hold almost all launches, set-up almost all the time
busy, idle, cur_end = 0, 0, ev[0][0]
gaps = collections.Counter()
gap_ns = collections.Counter()
last_name = None
for s, e, name in ev:
    if s > cur_end:
        idle += s - cur_end
        if last_name is not None:
            gaps[last_name[:80]] += 1
            gap_ns[last_name[:80]] += s - cur_end
        busy += e - s
        cur_end = e
        last_name = name
    else:
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
            last_name = name
span = ev[-1][1] - ev[0][0]
print(f"window {span / 1e6:.1f} ms, {len(ev)} launches: busy {busy / 1e6:.1f} ms ({100 * busy / span:.1f} %), idle {idle / 1e6:.1f} ms ({100 * idle / span:.1f} %)")
print("idle time by the kernel that ended before the gap:")
for name, ns in gap_ns.most_common(12):
    print(f"  {ns / 1e6:8.2f} ms in {gaps[name]:5d} gaps (avg {ns / gaps[name] / 1e3:6.1f} us)  after {name}")
