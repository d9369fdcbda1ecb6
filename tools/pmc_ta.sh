#!/bin/bash
# usage: tools/pmc_ta.sh <tag> [bench args...]  -- texture-path counters per kernel (own pass, kernel-trace only)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && export JT_NO_AUX=1
rocprofv3 --pmc TA_BUSY_avr TA_BUSY_max TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ta_$tag -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-probe --no-torch-baseline --no-extras "$@" > $GRAFT_REPO_ROOT/gpurun_out/ta_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY' gpurun_out/ta_$tag/k_counter_collection.csv
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); calls = defaultdict(set); dur = defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Kernel_Name"].replace("void ", "").split("(")[0][:44]
    if "jt::" not in nm: continue
    acc[nm][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in calls[nm]:
        calls[nm].add(r["Dispatch_Id"]); dur[nm] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
cols = ["TA_BUSY_avr", "TA_BUSY_max", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "GRBM_GUI_ACTIVE"]
print("%-46s %5s %8s " % ("kernel", "calls", "avg_us") + " ".join("%16s" % c[:16] for c in cols))
for nm in sorted(acc, key=lambda k: -dur[k])[:8]:
    n = len(calls[nm])
    print("%-46s %5d %8.1f " % (nm, n, dur[nm] / n) + " ".join("%16.0f" % (acc[nm][c] / n) for c in cols))
PY
