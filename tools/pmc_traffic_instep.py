#!/usr/bin/env python3
"""HBM-side traffic of k_shade_bwd / k_shade_fwd<train> INSIDE the training steps of bench.py, from two rocprofv3 --pmc
passes over the same seeded command (tools/round2_profiles.sh):

  python tools/pmc_traffic_instep.py FETCH/p_counter_collection.csv WRITE/p_counter_collection.csv bench_line.json out.json

FETCH_SIZE / WRITE_SIZE are in KiB.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 tallies the 128-byte
requests of wide (16 B / lane) reads at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16 B / lane stores and float
atomics.  The counters are averaged over EVERY launch of the process (priming and warm-up included) and divided by the
average shaded samples of those same launches, which bench.py reports (roofline.process_samples_per_launch)."""
import csv
import json
import sys
from collections import defaultdict

KERNELS = {"k_shade_bwd": "jt::k_shade_bwd<", "k_shade_fwd_train": "jt::k_shade_fwd<"}


def per_launch(path):
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        for key, pat in KERNELS.items():
            if pat in r["Kernel_Name"]:
                a = acc[key]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
                a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    return {k: (v[1] / v[0] * 1024.0, v[2] / v[0], v[0]) for k, v in acc.items()}


def main(fetch_csv, write_csv, bench_json, out):
    f, w = per_launch(fetch_csv), per_launch(write_csv)
    line = [ln for ln in open(bench_json).read().splitlines() if ln.startswith("{")][-1]
    roof = json.loads(line)["roofline"]
    n = roof["process_samples_per_launch"]
    res = {"note": " ".join(__doc__.split("\n\n")[-1].split()), "process_samples_per_launch": n,
           "process_launches": roof["process_launches"]}
    for key in KERNELS:
        if key not in f or key not in w:
            continue
        fetch, write = 2.0 * f[key][0], w[key][0]
        res[key] = {"launches_averaged": f[key][2], "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                    "hbm_bytes_per_launch": fetch + write, "hbm_bytes_per_sample": (fetch + write) / n,
                    "avg_us_under_pmc": f[key][1]}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
