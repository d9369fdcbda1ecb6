#!/usr/bin/env python3
"""Per-kernel stats from a rocprofv3 kernel_trace.csv, ignoring launches shorter than a threshold
(the chunked shade backward issues fixed-count launches; chunks beyond the shaded samples exit at once)."""
import csv, sys, collections
path = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0   # us
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 1
d = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for k, v in d.items():
    live = [x for x in v if x >= thr]
    rows.append((sum(v), k, len(v), len(live), sum(live) / max(len(live), 1), sum(v) - sum(live)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total %.3f ms (%.3f ms/iter)" % (tot / 1e3, tot / 1e3 / iters))
print("%-60s %6s %6s %10s %10s %9s" % ("kernel", "calls", "live", "avg_live_us", "total_ms", "ms/iter"))
for t, k, n, nl, avg, small in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 16]:
    print("%-60s %6d %6d %10.1f %10.3f %9.3f" % (k[:60], n, nl, avg, t / 1e3, t / 1e3 / iters))
