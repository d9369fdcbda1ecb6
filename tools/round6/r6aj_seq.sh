#!/bin/bash
# round 6, run aj: ordered kernel sequence of ONE eager headline iteration (start offsets, durations, queue)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/r6aj_trace -o k -- python3 $R/bench.py --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --no-roofline --steps 12 --warmup 3 > $O/r6aj.log 2>&1
ls $O/r6aj_trace
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/r6aj_trace/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
for f in glob.glob("$O/r6aj_trace/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), "copy"))
rows.sort()
cuts = [i for i, r in enumerate(rows) if "k_pose_fwd" in r[2]]
a, b = cuts[-4], cuts[-3]
t0 = rows[a][0]
prev = {}
lastend = None
for s, e, n, q in rows[a:b + 1]:
    gap = (s - lastend) / 1e3 if lastend else 0.0
    print("%9.1f  %7.1f us  q%-3s gap_since_any_end %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, gap, n[:100]))
    lastend = max(lastend or 0, e)
print("iteration period us:", (rows[b][0] - rows[a][0]) / 1e3)
PY
rm -rf $O/r6aj_trace
