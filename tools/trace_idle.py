#!/usr/bin/env python3
"""Device busy / idle accounting from a rocprofv3 kernel_trace.csv (--kernel-trace --output-format csv): the union of kernel intervals over
all queues, the idle gaps between them, and who follows the long gaps.  Usage: python tools/trace_idle.py <kernel_trace.csv> [skip_fraction]
(skip_fraction: leading share of the trace to ignore -- warm-up, default 0.5)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]) for r in rows), key=lambda e: e[0])
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t0 + (t1 - t0) * skip
ev = [e for e in ev if e[0] >= cut]
span = max(e[1] for e in ev) - ev[0][0]
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]; gaps = []
for s, e, n in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
ksum = sum(e - s for s, e, _ in ev)
print("window %.2f ms: %d launches, kernel time %.2f ms, device busy (union) %.2f ms = %.1f %%, idle %.2f ms in %d gaps" % (
    span / 1e6, len(ev), ksum / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(gaps)))
for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e12)):
    g = [x for x, _ in gaps if lo <= x < hi]
    print("  gaps %6.0f..%-8.0f ns: %5d, %.2f ms" % (lo, hi, len(g), sum(g) / 1e6))
from collections import Counter
c = Counter(); tt = Counter()
for x, n in gaps:
    if x >= 5e3: c[n] += 1; tt[n] += x
print("  kernels that follow gaps >= 5 us (count, total idle ms):")
for n, k in sorted(tt.items(), key=lambda kv: -kv[1])[:14]: print("   %4d %.2f  %s" % (c[n], k / 1e6, n))
