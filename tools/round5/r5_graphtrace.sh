#!/bin/bash
# Does the captured fork (JT_GRAPH_AUX=1) overlap inside a replayed hipGraph?  Kernel trace of the LLFF final-grid bench, replayed,
# then the overlap of the weight-gradient GEMMs with k_shade_scatter read off the timestamps.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="$R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-probe --no-torch-baseline --no-extras --config bat_llff_VM_MLP"
for mode in aux noaux eager; do
  case $mode in aux) export JT_GRAPH=1 JT_GRAPH_AUX=1;; noaux) export JT_GRAPH=1 JT_GRAPH_AUX=0;; eager) export JT_GRAPH=0 JT_GRAPH_AUX=0;; esac
  rm -rf $O/gt_$mode
  rocprofv3 --kernel-trace --output-format csv -d $O/gt_$mode -o k -- python3 $CMD > $O/gt_$mode.log 2>&1
  python3 - $O/gt_$mode <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id"), r.get("Stream_Id")))
rows.sort()
# the last iteration: from the last k_pose_fwd on
last = max(i for i, r in enumerate(rows) if "k_pose_fwd" in r[2])
prev = max(i for i, r in enumerate(rows[:last]) if "k_pose_fwd" in r[2])
it = rows[prev:last]
t0 = it[0][0]
print(sys.argv[1].split("/")[-1], "launches", len(it), "span us %.1f" % ((it[-1][1] - t0) / 1e3))
busy = sum(e - s for s, e, *_ in it) / 1e3
print("  sum of kernel durations us %.1f" % busy)
for s, e, n, q, st in it:
    print("  %8.1f %8.1f  q%s s%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, st, n))
PY
done > $O/r5_graphtrace.txt 2>&1
tail -5 $O/gt_aux.log
