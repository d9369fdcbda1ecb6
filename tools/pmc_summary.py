#!/usr/bin/env python3
"""Per-kernel averages of one rocprofv3 --pmc pass (counter_collection.csv): counter value and duration per launch.

usage: pmc_summary.py <counter_collection.csv> [name-substring ...]"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:60]


def main(path, subs):
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    counter = None
    for r in csv.DictReader(open(path)):
        nm = short(r["Kernel_Name"])
        if subs and not any(s in nm for s in subs):
            continue
        counter = r["Counter_Name"]
        a = acc[nm]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    print("%-62s %6s %16s %10s" % ("kernel", "calls", "avg " + str(counter), "avg_us"))
    for nm, (n, v, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print("%-62s %6d %16.1f %10.1f" % (nm, n, v / n, t / n))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
