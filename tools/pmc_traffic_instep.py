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

KERNELS = {"k_shade_bwd": "jt::k_shade_bwd<", "k_shade_fwd_train": "jt::k_shade_fwd<",
           "k_march_bwd_walk": "jt::k_march_bwd_walk<", "k_march_bwd_scan": "jt::k_march_bwd_scan", "k_march_fwd": "jt::k_march_fwd"}


# the appearance backward is ONE kernel (fused) or two launches per chunk (chain, then scatter: the default since round 5):
# "k_shade_bwd" stands for the per-sample backward as a whole -- the averages of the two kernels are added
EXTRA = {"k_shade_bwd": ["jt::k_shade_scatter<"]}


def per_launch(path):
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        for key, pat in KERNELS.items():
            if pat in r["Kernel_Name"]:
                a = acc[key]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
                a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        for key, pats in EXTRA.items():
            for pat in pats:
                if pat in r["Kernel_Name"]:
                    a = acc[(key, pat)]
                    a[0] += 1
                    a[1] += float(r["Counter_Value"])
                    a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    out = {k: (v[1] / v[0] * 1024.0, v[2] / v[0], v[0]) for k, v in acc.items() if not isinstance(k, tuple)}
    for k, v in acc.items():
        if isinstance(k, tuple) and k[0] in out:
            b = out[k[0]]
            out[k[0]] = (b[0] + v[1] / v[0] * 1024.0, b[1] + v[2] / v[0], b[2])
    return out


def main(fetch_csv, write_csv, bench_json, out):
    f, w = per_launch(fetch_csv), per_launch(write_csv)
    line = [ln for ln in open(bench_json).read().splitlines() if ln.startswith("{")][-1]
    roof = json.loads(line)["roofline"]
    n = roof["process_samples_per_launch"]
    res = {"note": " ".join(__doc__.split("\n\n")[-1].split()), "process_samples_per_launch": n,
           "process_launches": roof["process_launches"]}
    # the density kernels: per LISTED sample of the walk (in-box samples with a density gradient) when the bench line
    # counted them (roofline.density_backward: the timed launches' average), else per launch only
    db = roof.get("density_backward") or {}
    n_listed = db.get("samples_per_launch")
    res["density_backward_listed_samples_per_launch"] = n_listed
    for key in KERNELS:
        if key not in f or key not in w:
            continue
        fetch, write = 2.0 * f[key][0], w[key][0]
        res[key] = {"launches_averaged": f[key][2], "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                    "hbm_bytes_per_launch": fetch + write, "avg_us_under_pmc": f[key][1]}
        if key.startswith("k_shade"):
            res[key]["hbm_bytes_per_sample"] = (fetch + write) / n
            alg = roof["bytes_per_sample"] / 2 * (2 if key == "k_shade_bwd" else 1)
            res[key]["algorithmic_bytes_per_sample"] = alg
            res[key]["hbm_over_algorithmic"] = (fetch + write) / n / alg
        elif n_listed and key != "k_march_fwd":
            res[key]["hbm_bytes_per_listed_sample"] = (fetch + write) / n_listed
    if n_listed and "k_march_bwd_walk" in res and "k_march_bwd_scan" in res:
        tot = res["k_march_bwd_walk"]["hbm_bytes_per_launch"] + res["k_march_bwd_scan"]["hbm_bytes_per_launch"]
        res["density_backward"] = {"hbm_bytes_per_launch": tot, "hbm_bytes_per_listed_sample": tot / n_listed,
                                   "algorithmic_bytes_per_listed_sample": db.get("bytes_per_sample"),
                                   "hbm_over_algorithmic": tot / n_listed / db["bytes_per_sample"] if db.get("bytes_per_sample") else None}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
