#!/bin/bash
# usage: tools/pmc_sq3.sh <tag> [bench args...]  -- matrix-pipe counters per kernel (own pass, kernel-trace only)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && export JT_NO_AUX=1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sq3_$tag -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-probe --no-torch-baseline --no-extras "$@" > $GRAFT_REPO_ROOT/gpurun_out/sq3_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY' gpurun_out/sq3_$tag/k_counter_collection.csv
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); calls = defaultdict(set); dur = defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Kernel_Name"].replace("void ", "").split("(")[0][:44]
    if "jt::" not in nm: continue
    acc[nm][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in calls[nm]:
        calls[nm].add(r["Dispatch_Id"]); dur[nm] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
cols = ["SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F32"]
print("%-46s %5s %8s " % ("kernel", "calls", "avg_us") + " ".join("%14s" % c[3:][:14] for c in cols))
for nm in sorted(acc, key=lambda k: -dur[k])[:10]:
    n = len(calls[nm])
    print("%-46s %5d %8.1f " % (nm, n, dur[nm] / n) + " ".join("%14.0f" % (acc[nm][c] / n) for c in cols))
PY
