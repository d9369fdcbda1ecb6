#!/bin/bash
# ordered kernel / memcpy list of the last timed iterations of the default step
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
JT_NO_AUX=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/r4_order -o k -- python3 $R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --steps 3 --warmup 2 $* > $O/r4_order.log 2>&1
python3 - <<PY
import csv,glob
rows=[]
for f in glob.glob("$O/r4_order/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
for f in glob.glob("$O/r4_order/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY "+r.get("Direction","")+" "+r.get("Name","")))
rows.sort()
# last iteration = from the last k_pose_fwd... print the tail 75 entries
names=[r[2] for r in rows]
idx=[i for i,n in enumerate(names) if "k_pose_fwd" in n]   # an iteration starts with the pose composition
start=idx[-1] if idx else 0
prev=None
for s,e,n in rows[start:]:
    gap = (s-prev)/1000 if prev else 0
    print("%8.1f us  gap %6.1f  %s" % ((e-s)/1000, gap, n))
    prev=e
PY
rm -rf $O/r4_order
