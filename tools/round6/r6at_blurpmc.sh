#!/bin/bash
# round 6, run at: what holds the blur passes at 0.4 of the streaming roof?  SQ counters and HBM-side bytes of k_blur_mfma
cd $GRAFT_REPO_ROOT
bash tools/pmc_sq.sh r6at_blur --it 9000 > gpurun_out/r6at_sq.txt 2>&1
grep -E 'kernel|k_blur' gpurun_out/r6at_sq.txt
cd /tmp && export TMPDIR=/tmp && export JT_NO_AUX=1
for set in "FETCH_SIZE WRITE_SIZE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6at_$tag -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-probe --no-torch-baseline --no-extras --it 9000 > $GRAFT_REPO_ROOT/gpurun_out/r6at_$tag.log 2>&1
  python3 - "$GRAFT_REPO_ROOT/gpurun_out/r6at_$tag/k_counter_collection.csv" <<'PY'
import csv, sys, os
from collections import defaultdict
if not os.path.exists(sys.argv[1]):
    print("no counters:", sys.argv[1]); sys.exit(0)
acc = defaultdict(lambda: defaultdict(float)); calls = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Kernel_Name"].replace("void ", "").split("(")[0][:30]
    if "k_blur" not in nm: continue
    acc[nm][r["Counter_Name"]] += float(r["Counter_Value"]); calls[nm].add(r["Dispatch_Id"])
for nm in acc:
    n = len(calls[nm]); print(nm, n, {k: round(v / n, 1) for k, v in acc[nm].items()})
PY
done
tail -3 $GRAFT_REPO_ROOT/gpurun_out/r6at_TCC_HIT_sum_TCC_MISS_sum_TCC_R.log
