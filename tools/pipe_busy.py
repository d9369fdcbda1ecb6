#!/usr/bin/env python3
"""Per-launch busy fractions of the step's big kernels from the two counter passes of tools/pmc_sq3.sh and tools/pmc_ta.sh
(one rocprofv3 --pmc pass each, kernel trace only) -> the JSON bench.py reports as roofline.pipe_busy.

  python tools/pipe_busy.py <sq3 counter csv> <ta counter csv> <kernel_stats csv of a trace of the same command> <iterations> <out.json>

matrix pipes = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles); vector instructions = 4 x SQ_ACTIVE_INST_VALU /
(1024 x kernel cycles); texture path = TA_BUSY_avr / (GRBM_GUI_ACTIVE / 8); kernel cycles = GRBM_GUI_ACTIVE / 8 of the texture pass."""
import csv
import json
import sys
from collections import defaultdict


def load(path):
    acc, calls, dur = defaultdict(lambda: defaultdict(float)), defaultdict(set), defaultdict(float)
    for r in csv.DictReader(open(path)):
        nm = r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]
        if "jt::" not in nm:
            continue
        acc[nm][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in calls[nm]:
            calls[nm].add(r["Dispatch_Id"])
            dur[nm] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    return acc, {k: len(v) for k, v in calls.items()}, dur


def main():
    sq, ta, stats, iters, out = sys.argv[1:6]
    a, na, da = load(sq)
    b, nb, db = load(ta)
    kernels = {}
    for nm in sorted(da, key=lambda k: -da[k])[:8]:
        if nm not in b or da[nm] / na[nm] < 40.0:
            continue
        cyc = b[nm]["GRBM_GUI_ACTIVE"] / nb[nm] / 8.0
        kernels[nm] = {"avg_us_under_pmc": round(da[nm] / na[nm], 1),
                       "mfma_busy_frac": round(a[nm]["SQ_VALU_MFMA_BUSY_CYCLES"] / na[nm] / (1024.0 * cyc), 3),
                       "valu_active_frac": round(4.0 * a[nm]["SQ_ACTIVE_INST_VALU"] / na[nm] / (1024.0 * cyc), 3),
                       "texture_path_busy_frac": round(b[nm]["TA_BUSY_avr"] / nb[nm] / cyc, 3)}
    launches = sum(int(r["Calls"]) for r in csv.DictReader(open(stats)))
    json.dump({"note": __doc__.split("\n\n")[-1].replace("\n", " "), "kernels": kernels,
               "launches_per_step_profiled": round(launches / float(iters), 1)}, open(out, "w"), indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
