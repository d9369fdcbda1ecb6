#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats csv (kernel_stats.csv) into a short text table."""
import csv
import sys


def main(path, n=25, iters=None):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel time %.3f ms over %d kernels%s" % (tot / 1e6, len(rows), "" if not iters else " (%.3f ms/iter over %d iters)" % (tot / 1e6 / iters, iters)))
    print("%-72s %7s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "%"))
    for r in rows[:n]:
        print("%-72s %7s %10.3f %10.1f %6.1f" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                 float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25, int(sys.argv[3]) if len(sys.argv) > 3 else None)
