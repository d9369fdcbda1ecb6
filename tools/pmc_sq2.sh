#!/bin/bash
# usage: tools/pmc_sq2.sh <tag> [bench args...]  -- second set of SQ counters per kernel (LDS / VMEM pressure)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && export JT_NO_AUX=1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sq2_$tag -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-probe --no-torch-baseline --no-extras "$@" > $GRAFT_REPO_ROOT/gpurun_out/sq2_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' gpurun_out/sq2_$tag/k_counter_collection.csv
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); calls = defaultdict(set); dur = defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Kernel_Name"].replace("void ", "").split("(")[0][:40]
    if "jt::" not in nm: continue
    acc[nm][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in calls[nm]:
        calls[nm].add(r["Dispatch_Id"]); dur[nm] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
cols = ["SQ_INSTS_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_INSTS_BRANCH"]
print("%-42s %5s %8s " % ("kernel", "calls", "avg_us") + " ".join("%13s" % c[3:][:13] for c in cols))
for nm in sorted(acc, key=lambda k: -dur[k])[:6]:
    n = len(calls[nm])
    print("%-42s %5d %8.1f " % (nm, n, dur[nm] / n) + " ".join("%13.0f" % (acc[nm][c] / n) for c in cols))
PY
