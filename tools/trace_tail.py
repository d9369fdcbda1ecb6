#!/usr/bin/env python3
"""Per-kernel totals over the LAST part of a rocprofv3 kernel_trace.csv (by start time): what an iteration of the final stage of
a training run costs, stage by stage being one process.   tools/trace_tail.py <kernel_trace.csv> <fraction 0..1> <iterations in
that part, 0 = count the k_adam_batch launches> [rows]"""
import collections
import csv
import sys

path, frac, iters = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
nrow = int(sys.argv[4]) if len(sys.argv) > 4 else 30
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
cut = t1 - frac * (t1 - t0)
d = collections.defaultdict(list)
for s, e, k in rows:
    if s >= cut:
        d[k].append((e - s) / 1e3)
tab = sorted(((sum(v), k, len(v)) for k, v in d.items()), reverse=True)
tot = sum(t for t, _, _ in tab)
if iters <= 0:  # count the iterations by the optimizer launches (one k_adam_batch per iteration)
    iters = max(1, sum(n for _, k, n in tab if "k_adam_batch" in k))
print("last %.0f %% of the trace: %.2f s wall, kernel time %.3f s = %.3f ms/iter over %d iterations (%.1f launches/iter)"
      % (100 * frac, (t1 - cut) / 1e9, tot / 1e6, tot / 1e3 / iters, iters, sum(n for _, _, n in tab) / iters))
print("%-78s %8s %10s %9s %6s" % ("kernel", "calls", "avg_us", "ms/iter", "%"))
for t, k, n in tab[:nrow]:
    print("%-78s %8d %10.1f %9.4f %6.1f" % (k[:78], n, t / n, t / 1e3 / iters, 100 * t / tot))
